"""Raw (non-autograd) calls into the HIP library on torch device tensors.

PyTorch is used here for device memory and the current HIP stream only; every computation below is a
kernel of libefm_hip.so.  Activations are NHWC tensors (B, H, W, Cp) with Cp = pad4(C) and zero pad
channels; 2-D (rows, Cp) tensors are the H = W = 1 case.
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check, conv_desc, pad4, pad16  # noqa: F401

_workspace = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.EfmError("efm ops need device tensors (the HIP library is the only compute path)")
        if t.dtype not in (torch.float32, torch.int32, torch.bfloat16, torch.uint8):
            raise _lib.EfmError("efm ops take float32 / bfloat16 / int32 / uint8 tensors, got %s" % t.dtype)
        if not t.is_contiguous():
            raise _lib.EfmError("efm ops need contiguous tensors")


def workspace(nbytes, device):
    """Grow-only per-device scratch buffer (split-K slabs); never allocated inside a captured region twice."""
    key = (device.type, device.index)
    ws = _workspace.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            torch.cuda.synchronize(device)  # kernels on any stream may still use the old buffer (first steps only)
        ws = torch.empty((nbytes + 3) // 4 + 1024, dtype=torch.float32, device=device)
        _workspace[key] = ws
    return ws


# ------------------------------------------------------------------------------------------ conv
def conv_weight_shape(d):
    return (d.n_pad16, d.k_pad)


def conv_pack_weights(d, w_oihw):
    _need_dev(w_oihw)
    wp = torch.empty(conv_weight_shape(d), dtype=torch.float32, device=w_oihw.device)
    check(_lib.load().efm_conv_pack_weights(ctypes.byref(d), _p(w_oihw), _p(wp), _stream()), "efm_conv_pack_weights")
    return wp


def conv_pack_weights_into(d, w_oihw, wp):
    _need_dev(w_oihw, wp)
    check(_lib.load().efm_conv_pack_weights(ctypes.byref(d), _p(w_oihw), _p(wp), _stream()), "efm_conv_pack_weights")


def conv_unpack_weights(d, wp):
    _need_dev(wp)
    w = torch.empty((d.cout, d.cin, d.kh, d.kw), dtype=torch.float32, device=wp.device)
    check(_lib.load().efm_conv_unpack_weights(ctypes.byref(d), _p(wp), _p(w), _stream()), "efm_conv_unpack_weights")
    return w


def conv_make_dgrad_weights(d, wp, out=None):
    _need_dev(wp, out)
    if out is None:
        out = torch.empty((d.dn_pad16, d.dk_pad), dtype=torch.float32, device=wp.device)
    check(_lib.load().efm_conv_make_dgrad_weights(ctypes.byref(d), _p(wp), _p(out), _stream()), "efm_conv_make_dgrad_weights")
    return out


def conv_fwd(d, x, wp, bias=None, residual=None, out=None):
    _need_dev(x, wp, bias, residual, out)
    if out is None:
        out = torch.empty((d.batch, d.hout, d.wout, d.cout_p), dtype=torch.float32, device=x.device)
    assert x.numel() == d.batch * d.hin * d.win * d.cin_p, (x.shape, d.batch, d.hin, d.win, d.cin_p)
    assert out.numel() == d.batch * d.hout * d.wout * d.cout_p
    assert wp.numel() == d.n_pad16 * d.k_pad
    assert bias is None or bias.numel() == d.n_pad16
    assert residual is None or residual.numel() == out.numel()
    check(_lib.load().efm_conv_fwd(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(residual), _p(out), _stream()), "efm_conv_fwd")
    return out


def conv_mfm_supported(d):
    return bool(_lib.load().efm_conv_mfm_supported(ctypes.byref(d)))


def conv_mfm_fwd(d, x, wp, bias, ways=3, order=_lib.MFM_ORDER_GROUP, pool=False):
    """Fused conv + bias + MFM (+ 2x2 max pooling): returns (z, route)."""
    _need_dev(x, wp, bias)
    assert x.numel() == d.batch * d.hin * d.win * d.cin_p and wp.numel() == d.n_pad16 * d.k_pad
    co = pad4(mfm_out_channels(d.cout, ways))
    ho, wo = (d.hout // 2, d.wout // 2) if pool else (d.hout, d.wout)
    z = torch.empty((d.batch, ho, wo, co), dtype=torch.float32, device=x.device)
    route = torch.empty((d.batch, ho, wo, co), dtype=torch.uint8, device=x.device)
    check(_lib.load().efm_conv_mfm_fwd(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(z), ctypes.c_void_p(route.data_ptr()),
                                       ways, order, int(bool(pool)), _stream()), "efm_conv_mfm_fwd")
    return z, route


def mfm_pool_bwd(d, route, dz, ways=3, pool=False):
    """Gradient of the fused epilogue w.r.t. the (never materialised) conv output: (B, hout, wout, cout_p)."""
    _need_dev(dz)
    dy = torch.empty((d.batch, d.hout, d.wout, d.cout_p), dtype=torch.float32, device=dz.device)
    check(_lib.load().efm_mfm_pool_bwd(ctypes.c_void_p(route.data_ptr()), _p(dz), _p(dy), d.batch, d.hout, d.wout, d.cout, ways,
                                       int(bool(pool)), _stream()), "efm_mfm_pool_bwd")
    return dy


def conv_bwd_data(d, dy, wd, add=None, out=None):
    _need_dev(dy, wd, add, out)
    if out is None:
        out = torch.empty((d.batch, d.hin, d.win, d.cin_p), dtype=torch.float32, device=dy.device)
    assert dy.numel() == d.batch * d.hout * d.wout * d.cout_p
    assert out.numel() == d.batch * d.hin * d.win * d.cin_p
    assert wd.numel() == d.dn_pad16 * d.dk_pad
    assert add is None or add.numel() == out.numel()
    check(_lib.load().efm_conv_bwd_data(ctypes.byref(d), _p(dy), _p(wd), _p(add), _p(out), _stream()), "efm_conv_bwd_data")
    return out


def conv_bwd_weight(d, x, dy, dw=None, dbias=None, want_bias=True, accumulate=False):
    _need_dev(x, dy, dw, dbias)
    if dw is None:
        dw = torch.empty(conv_weight_shape(d), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((d.n_pad16,), dtype=torch.float32, device=x.device)
    assert x.numel() == d.batch * d.hin * d.win * d.cin_p
    assert dy.numel() == d.batch * d.hout * d.wout * d.cout_p
    assert dw.numel() == d.n_pad16 * d.k_pad
    lib = _lib.load()
    nbytes = lib.efm_conv_wgrad_workspace_bytes(ctypes.byref(d))
    ws = workspace(nbytes, x.device)
    check(lib.efm_conv_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(dbias if want_bias else None), int(bool(accumulate)), _p(ws),
                                  ctypes.c_size_t(ws.numel() * 4), _stream()), "efm_conv_bwd_weight")
    return dw, (dbias if want_bias else None)


def conv_wgrad_workspace_bytes(d):
    return int(_lib.load().efm_conv_wgrad_workspace_bytes(ctypes.byref(d)))


def conv_bwd_weight_slabs(d, x, dy, ws, want_bias=True):
    """First launch of the weight gradient (matrix-core kernel -> slabs in the caller's workspace `ws`) on the current stream."""
    _need_dev(x, dy, ws)
    check(_lib.load().efm_conv_bwd_weight_slabs(ctypes.byref(d), _p(x), _p(dy), int(bool(want_bias)), _p(ws), ctypes.c_size_t(ws.numel() * 4),
                                                _stream()), "efm_conv_bwd_weight_slabs")


def conv_bwd_weight_finish(d, ws, dw, dbias=None, accumulate=False):
    """Second launch (fixed-order reduction of the slabs in `ws` into dw / dbias) on the current stream."""
    _need_dev(ws, dw, dbias)
    check(_lib.load().efm_conv_bwd_weight_finish(ctypes.byref(d), _p(dw), _p(dbias), int(bool(accumulate)), _p(ws),
                                                 ctypes.c_size_t(ws.numel() * 4), _stream()), "efm_conv_bwd_weight_finish")


# ------------------------------------------------------------------------------------ elementwise
def nchw_to_nhwc(x, out=None):
    _need_dev(x, out)
    b, c, h, w = x.shape
    if out is None:
        out = torch.empty((b, h, w, pad4(c)), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_nchw_to_nhwc(_p(x), _p(out), b, c, h, w, _stream()), "efm_nchw_to_nhwc")
    return out


def rowpack_nchw(x, kw, pad_w, bf16=False):
    """NCHW fp32 -> row-packed NHWC (efm_rowpack_nchw): (B, H, W, pad(kw*C)), fp32 (channel stride pad4) or bf16 (pad8)."""
    _need_dev(x)
    b, c, h, w = x.shape
    cp = (kw * c + 7) & ~7 if bf16 else pad4(kw * c)
    out = torch.empty((b, h, w, cp), dtype=torch.bfloat16 if bf16 else torch.float32, device=x.device)
    check(_lib.load().efm_rowpack_nchw(_p(x), _p(out), b, c, h, w, kw, pad_w, 1 if bf16 else 0, _stream()), "efm_rowpack_nchw")
    return out


def crop_mirror_u8(src, crop, h, w, scale=1.0):
    """ImageRecordIter's crop / mirror / scale on the device: src uint8 (B, IH, IW, C) device tensor, crop int32 (B, 3) = (y0, x0,
    mirror) device tensor -> fp32 NCHW (B, C, h, w)."""
    _need_dev(crop)
    if src.dtype != torch.uint8 or not src.is_cuda or not src.is_contiguous():
        raise _lib.EfmError("crop_mirror_u8 needs a contiguous uint8 device tensor")
    b, ih, iw, c = src.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=src.device)
    check(_lib.load().efm_crop_mirror_u8(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(crop.data_ptr()), _p(out), b, ih, iw, c, h, w,
                                         float(scale), _stream()), "efm_crop_mirror_u8")
    return out


def nhwc_to_nchw(x, c, out=None):
    _need_dev(x, out)
    b, h, w, cp = x.shape
    assert cp == pad4(c)
    if out is None:
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_nhwc_to_nchw(_p(x), _p(out), b, c, h, w, _stream()), "efm_nhwc_to_nchw")
    return out


def mfm_out_channels(c, ways):
    return 2 * c // 3 if ways == 3 else c // 2


def mfm_fwd(x, c, ways=3, out=None):
    _need_dev(x, out)
    assert x.shape[-1] == pad4(c)
    rows = x.numel() // x.shape[-1]
    if out is None:
        out = torch.empty(x.shape[:-1] + (pad4(mfm_out_channels(c, ways)),), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_mfm_fwd(_p(x), _p(out), rows, c, ways, _stream()), "efm_mfm_fwd")
    return out


def mfm_bwd(x, dy, c, ways=3, order=_lib.MFM_ORDER_GROUP, add=None, out=None):
    _need_dev(x, dy, add, out)
    rows = x.numel() // x.shape[-1]
    assert dy.numel() == rows * pad4(mfm_out_channels(c, ways))
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().efm_mfm_bwd(_p(x), _p(dy), _p(add), _p(out), rows, c, ways, order, _stream()), "efm_mfm_bwd")
    return out


def maxpool2_fwd(x, c, out=None):
    _need_dev(x, out)
    b, h, w, cp = x.shape
    assert cp == pad4(c)
    if out is None:
        out = torch.empty((b, h // 2, w // 2, cp), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_maxpool2_fwd(_p(x), _p(out), b, h, w, c, _stream()), "efm_maxpool2_fwd")
    return out


def maxpool2_bwd(x, dy, c, out=None):
    _need_dev(x, dy, out)
    b, h, w, cp = x.shape
    assert dy.numel() == b * (h // 2) * (w // 2) * cp
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().efm_maxpool2_bwd(_p(x), _p(dy), _p(out), b, h, w, c, _stream()), "efm_maxpool2_bwd")
    return out


# ------------------------------------------------------------------------------------------ head
def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def _need_rows(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise _lib.EfmError("head ops take 2-D float32 device tensors with unit inner stride")


def l2norm_fwd(x, mode=_lib.L2_ROW):
    _need_rows(x)
    rows, d = x.shape
    y = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    norm = torch.empty((rows if mode == _lib.L2_ROW else 1,), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_l2norm_fwd(_p(x), _p(y), _p(norm), rows, d, _ld(x), d, mode, _stream()), "efm_l2norm_fwd")
    return y, norm


def l2norm_bwd(y, norm, dy, mode=_lib.L2_ROW):
    _need_rows(y, dy)
    rows, d = y.shape
    dx = torch.empty((rows, d), dtype=torch.float32, device=y.device)
    check(_lib.load().efm_l2norm_bwd(_p(y), _p(norm), _p(dy), _p(dx), rows, d, _ld(y), _ld(dy), d, mode, _stream()), "efm_l2norm_bwd")
    return dx


def gather_rows(x, idx):
    _need_rows(x)
    assert idx.dtype == torch.int32 and idx.is_cuda and idx.is_contiguous()
    rows, d = idx.numel(), x.shape[1]
    y = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_gather_rows(_p(x), _p(idx), _p(y), rows, d, _ld(x), d, _stream()), "efm_gather_rows")
    return y


def triplet_fwd(a, p, n, margin):
    _need_rows(a, p, n)
    rows, d = a.shape
    loss = torch.empty((rows,), dtype=torch.float32, device=a.device)
    check(_lib.load().efm_triplet_fwd(_p(a), _p(p), _p(n), _p(loss), rows, d, _ld(a), _ld(p), _ld(n), float(margin), _stream()), "efm_triplet_fwd")
    return loss


def triplet_bwd(a, p, n, loss, gloss, need_dn=False, da=None, dp=None, dn=None):
    """Gradients of the loss vector w.r.t. anchor / positive (/ negative).  `da`, `dp`, `dn` may be preallocated
    row blocks (unit inner stride, identical row stride) — e.g. the two halves of one (B, D) buffer."""
    _need_rows(a, p, n, da, dp, dn)
    rows, d = a.shape
    if da is None:
        da = torch.empty((rows, d), dtype=torch.float32, device=a.device)
    if dp is None:
        dp = torch.empty((rows, d), dtype=torch.float32, device=a.device)
    if dn is None and need_dn:
        dn = torch.empty((rows, d), dtype=torch.float32, device=a.device)
    ldg = _ld(da)
    assert _ld(dp) == ldg and (dn is None or _ld(dn) == ldg)
    check(_lib.load().efm_triplet_bwd(_p(a), _p(p), _p(n), _p(loss), _p(gloss), _p(da), _p(dp), _p(dn), rows, d,
                                      _ld(a), _ld(p), _ld(n), ldg, _stream()), "efm_triplet_bwd")
    return da, dp, dn


def triplet_indexed_fwd(e, pos, neg, margin):
    _need_rows(e)
    rows, d = e.shape
    loss = torch.empty((rows,), dtype=torch.float32, device=e.device)
    check(_lib.load().efm_triplet_indexed_fwd(_p(e), _p(pos), _p(neg), _p(loss), rows, d, _ld(e), float(margin), _stream()), "efm_triplet_indexed_fwd")
    return loss


def triplet_indexed_bwd(e, pos, neg, inv_pos, loss, gloss, out=None):
    _need_rows(e, out)
    rows, d = e.shape
    if out is None:
        out = torch.empty((rows, d), dtype=torch.float32, device=e.device)
    check(_lib.load().efm_triplet_indexed_bwd(_p(e), _p(pos), _p(neg), _p(inv_pos), _p(loss), _p(gloss), _p(out), rows, d, _ld(e), _ld(out),
                                              _stream()), "efm_triplet_indexed_bwd")
    return out


def cosine_pairs(a, p, n):
    _need_rows(a, p, n)
    rows, d = a.shape
    s_ap = torch.empty((rows,), dtype=torch.float32, device=a.device)
    s_an = torch.empty_like(s_ap)
    check(_lib.load().efm_cosine_pairs(_p(a), _p(p), _p(n), _p(s_ap), _p(s_an), rows, d, _ld(a), _ld(p), _ld(n), _stream()), "efm_cosine_pairs")
    return s_ap, s_an


def pair_distance(a, b, mean=None):
    """-> (squared Euclidean distance, cosine similarity) per row, optionally after subtracting `mean` from both."""
    _need_rows(a, b)
    rows, d = a.shape
    sq = torch.empty((rows,), dtype=torch.float32, device=a.device)
    cs = torch.empty_like(sq)
    check(_lib.load().efm_pair_distance(_p(a), _p(b), _p(mean), _p(sq), _p(cs), rows, d, _ld(a), _ld(b), _stream()), "efm_pair_distance")
    return sq, cs


def gallery_match(query, gallery, topk=1):
    """Scores of every query against every gallery row + the top-k matches (ref: Feature.hpp:345-392 keeps the best
    `sim_th`-passing rows).  Returns (scores (nq, n), top values (nq, k), top indices (nq, k))."""
    _need_rows(query, gallery)
    nq, d = query.shape
    n = gallery.shape[0]
    scores = torch.empty((nq, n), dtype=torch.float32, device=query.device)
    check(_lib.load().efm_gallery_scores(_p(query), _p(gallery), _p(scores), nq, n, d, _ld(query), _ld(gallery), _stream()), "efm_gallery_scores")
    vals, idx = torch.topk(scores, min(topk, n), dim=1)  # selection only; the arithmetic is the kernel's
    return scores, vals, idx


def gram_cosine(e):
    _need_rows(e)
    rows, d = e.shape
    g = torch.empty((rows, rows), dtype=torch.float32, device=e.device)
    check(_lib.load().efm_gram_cosine(_p(e), _p(g), rows, d, _ld(e), _stream()), "efm_gram_cosine")
    return g


def mine_semihard(g, labels, anchor_idx, pos_idx):
    for t in (labels, anchor_idx, pos_idx):
        assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous()
    n_anchor, rows = anchor_idx.numel(), g.shape[0]
    neg = torch.empty((n_anchor,), dtype=torch.int32, device=g.device)
    check(_lib.load().efm_mine_semihard(_p(g), _p(labels), _p(anchor_idx), _p(pos_idx), _p(neg), n_anchor, rows, _stream()), "efm_mine_semihard")
    return neg


# ------------------------------------------------------------------------------------- optimiser
def sgd_update(w, g, lr, wd=0.0, rescale=1.0):
    _need_dev(w, g)
    check(_lib.load().efm_sgd_update(_p(w), _p(g), w.numel(), float(lr), float(wd), float(rescale), _stream()), "efm_sgd_update")


def adam_update(w, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, rescale=1.0):
    _need_dev(w, g, m, v)
    check(_lib.load().efm_adam_update(_p(w), _p(g), _p(m), _p(v), w.numel(), float(lr), float(beta1), float(beta2),
                                      float(eps), float(wd), float(rescale), int(step), _stream()), "efm_adam_update")


# ------------------------------------------------------------------------------- bf16 tensor-core path
def pad8(c):
    return (c + 7) & ~7


def pad32(c):
    return (c + 31) & ~31


def _bf(t):
    assert t is None or t.dtype == torch.bfloat16, "bf16 path: expected a bfloat16 tensor"
    return t


def nchw_to_nhwc_bf16(x):
    _need_dev(x)
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, pad8(c)), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().efm_nchw_to_nhwc_bf16(_p(x), _p(out), b, c, h, w, _stream()), "efm_nchw_to_nhwc_bf16")
    return out


def convb_cast_weights(d, wp, wb=None, wdb=None, need_dgrad=True):
    """fp32 packed master weight -> (bf16 forward weight, bf16 data-gradient weight or None)."""
    _need_dev(wp, wb, wdb)
    lib = _lib.load()
    if wb is None:
        wb = torch.empty(lib.efm_convb_weight_elems(ctypes.byref(d)), dtype=torch.bfloat16, device=wp.device)
    if wdb is None and need_dgrad:
        wdb = torch.empty(lib.efm_convb_dgrad_weight_elems(ctypes.byref(d)), dtype=torch.bfloat16, device=wp.device)
    check(lib.efm_convb_cast_weights(ctypes.byref(d), _p(wp), _p(wb), _p(wdb if need_dgrad else None), _stream()), "efm_convb_cast_weights")
    return wb, (wdb if need_dgrad else None)


def convb_fwd(d, x, wb, bias=None, residual=None):
    _need_dev(_bf(x), _bf(wb), bias, _bf(residual))
    assert x.numel() == d.batch * d.hin * d.win * pad8(d.cin)
    y = torch.empty((d.batch, d.hout, d.wout, pad8(d.cout)), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().efm_convb_fwd(ctypes.byref(d), _p(x), _p(wb), _p(bias), _p(residual), _p(y), _stream()), "efm_convb_fwd")
    return y


def convb_mfm_fwd(d, x, wb, bias, ways=3, order=_lib.MFM_ORDER_GROUP, pool=False, out_f32=False):
    _need_dev(_bf(x), _bf(wb), bias)
    assert x.numel() == d.batch * d.hin * d.win * pad8(d.cin)
    co = mfm_out_channels(d.cout, ways)
    cpo = pad4(co) if out_f32 else pad8(co)
    ho, wo = (d.hout // 2, d.wout // 2) if pool else (d.hout, d.wout)
    z = torch.empty((d.batch, ho, wo, cpo), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    route = torch.empty((d.batch, ho, wo, cpo), dtype=torch.uint8, device=x.device)
    check(_lib.load().efm_convb_mfm_fwd(ctypes.byref(d), _p(x), _p(wb), _p(bias), _p(z), _p(route), ways, order, int(bool(pool)),
                                        int(bool(out_f32)), _stream()), "efm_convb_mfm_fwd")
    return z, route


def convb_bwd_data(d, dy, wdb, add=None):
    _need_dev(_bf(dy), _bf(wdb), _bf(add))
    assert dy.numel() == d.batch * d.hout * d.wout * pad8(d.cout)
    dx = torch.empty((d.batch, d.hin, d.win, pad8(d.cin)), dtype=torch.bfloat16, device=dy.device)
    check(_lib.load().efm_convb_bwd_data(ctypes.byref(d), _p(dy), _p(wdb), _p(add), _p(dx), _stream()), "efm_convb_bwd_data")
    return dx


def convb_mfm_pool_bwd(d, route, dz, ways=3, pool=False):
    _need_dev(dz, route)
    dy = torch.empty((d.batch, d.hout, d.wout, pad8(d.cout)), dtype=torch.bfloat16, device=dz.device)
    check(_lib.load().efm_convb_mfm_pool_bwd(_p(route), _p(dz), int(dz.dtype == torch.float32), _p(dy), d.batch, d.hout, d.wout, d.cout,
                                             ways, int(bool(pool)), _stream()), "efm_convb_mfm_pool_bwd")
    return dy


def convb_bwd_weight(d, x, dy, dw=None, dbias=None, want_bias=True, accumulate=False):
    _need_dev(_bf(x), _bf(dy), dw, dbias)
    if dw is None:
        dw = torch.empty(conv_weight_shape(d), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((d.n_pad16,), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = workspace(lib.efm_convb_wgrad_workspace_bytes(ctypes.byref(d)), x.device)
    check(lib.efm_convb_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(dbias if want_bias else None), int(bool(accumulate)), _p(ws),
                                   ctypes.c_size_t(ws.numel() * 4), _stream()), "efm_convb_bwd_weight")
    return dw, (dbias if want_bias else None)


def convb_mfm_bwd_weight_supported(d, ways, pool):
    return bool(_lib.load().efm_convb_mfm_bwd_weight_supported(ctypes.byref(d), int(ways), int(bool(pool))))


def convb_mfm_bwd_weight(d, x, route, dz, ways, pool, dw=None, dbias=None, want_bias=True, accumulate=False):
    """Weight (+ bias) gradient of a conv -> MFM2 -> 2x2 pooling layer straight from dz and the route bytes: the conv-output gradient is
    formed in LDS, never in HBM (efm_convb_mfm_bwd_weight; replaces convb_mfm_pool_bwd + convb_bwd_weight where the input needs no gradient)."""
    _need_dev(_bf(x), _bf(dz), route, dw, dbias)
    if dw is None:
        dw = torch.empty(conv_weight_shape(d), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((d.n_pad16,), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = workspace(lib.efm_convb_wgrad_workspace_bytes(ctypes.byref(d)), x.device)
    check(lib.efm_convb_mfm_bwd_weight(ctypes.byref(d), _p(x), _p(route), _p(dz), int(ways), int(bool(pool)), _p(dw), _p(dbias if want_bias else None),
                                       int(bool(accumulate)), _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()), "efm_convb_mfm_bwd_weight")
    return dw, (dbias if want_bias else None)


# ---- Winograd F(2x2, 3x3) form of the 3x3 / pad 1 convolutions ----------------------------------------------------------
PASS_FWD, PASS_FUSED, PASS_DGRAD, PASS_WGRAD, PASS_WINO_FWD, PASS_WINO_FUSED, PASS_WINO_DGRAD = range(7)


def conv_kernel_info(d, pass_, ways=0, pool=False):
    """-> (kernel instance name, matrix-core flops the launch EXECUTES) for roofline accounting (efm_conv_kernel_info)."""
    name = ctypes.create_string_buffer(96)
    fl = ctypes.c_double()
    check(_lib.load().efm_conv_kernel_info(ctypes.byref(d), int(pass_), int(ways), int(bool(pool)), name, 96, ctypes.byref(fl)),
          "efm_conv_kernel_info")
    return name.value.decode(), fl.value


def wino_supported(d):
    return bool(_lib.load().efm_wino_supported(ctypes.byref(d)))


def wino_make_u(d, w, dgrad=False, out=None):
    """Transformed weights U = G g G^T from the packed fp32 weights (dgrad: their tap-flipped transpose, for wino_bwd_data)."""
    _need_dev(w)
    n = _lib.load().efm_wino_u_elems(ctypes.byref(d), 1 if dgrad else 0)
    u = out if out is not None else torch.empty((n,), dtype=torch.float32, device=w.device)
    check(_lib.load().efm_wino_make_u(ctypes.byref(d), _p(w), _p(u), 1 if dgrad else 0, _stream()), "efm_wino_make_u")
    return u


def wino_u_numel(d, dgrad=False, ways=0):
    lib = _lib.load()
    return int(lib.efm_wino_mfm_u_elems(ctypes.byref(d), ways) if ways else lib.efm_wino_u_elems(ctypes.byref(d), 1 if dgrad else 0))


def wino_make_u_batch(jobs):
    """jobs: list of (desc, packed weights, u buffer, dgrad flag, ways) -> every U in one launch (efm_wino_make_u_batch)."""
    n = len(jobs)
    if not n:
        return
    descs = (ctypes.POINTER(_lib.ConvDesc) * n)(*[ctypes.pointer(j[0]) for j in jobs])
    ws = (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in jobs])
    us = (ctypes.c_void_p * n)(*[j[2].data_ptr() for j in jobs])
    dg = (ctypes.c_int * n)(*[1 if j[3] else 0 for j in jobs])
    wy = (ctypes.c_int * n)(*[int(j[4]) for j in jobs])
    check(_lib.load().efm_wino_make_u_batch(n, descs, ws, us, dg, wy, _stream()), "efm_wino_make_u_batch")


def wino_fwd(d, x, u, bias=None, residual=None, out=None):
    _need_dev(x, u)
    y = out if out is not None else torch.empty((d.batch, d.hout, d.wout, d.cout_p), dtype=torch.float32, device=x.device)
    check(_lib.load().efm_wino_fwd(ctypes.byref(d), _p(x), _p(u), _p(bias), _p(residual), _p(y), _stream()), "efm_wino_fwd")
    return y


def wino_bwd_data(d, dy, u_dgrad, add=None, out=None):
    _need_dev(dy, u_dgrad)
    dx = out if out is not None else torch.empty((d.batch, d.hin, d.win, d.cin_p), dtype=torch.float32, device=dy.device)
    check(_lib.load().efm_wino_bwd_data(ctypes.byref(d), _p(dy), _p(u_dgrad), _p(add), _p(dx), _stream()), "efm_wino_bwd_data")
    return dx


def wino_mfm_make_u(d, w, ways, out=None):
    _need_dev(w)
    n = _lib.load().efm_wino_mfm_u_elems(ctypes.byref(d), ways)
    u = out if out is not None else torch.empty((n,), dtype=torch.float32, device=w.device)
    check(_lib.load().efm_wino_mfm_make_u(ctypes.byref(d), _p(w), _p(u), ways, _stream()), "efm_wino_mfm_make_u")
    return u


def wino_mfm_fwd(d, x, u, bias, ways=3, order=_lib.MFM_ORDER_GROUP, pool=False):
    """Winograd form of conv_mfm_fwd: -> (z, route) in the same layout."""
    _need_dev(x, u)
    cs = d.cout // ways
    co = 2 * cs if ways == 3 else cs
    h, w = (d.hout // 2, d.wout // 2) if pool else (d.hout, d.wout)
    z = torch.empty((d.batch, h, w, pad4(co)), dtype=torch.float32, device=x.device)
    route = torch.empty((d.batch, h, w, pad4(co)), dtype=torch.uint8, device=x.device)
    check(_lib.load().efm_wino_mfm_fwd(ctypes.byref(d), _p(x), _p(u), _p(bias), _p(z), _p(route), ways, order, 1 if pool else 0, _stream()),
          "efm_wino_mfm_fwd")
    return z, route


def wino_bwd_weight(d, x, dy, dw=None, dbias=None, want_bias=True, accumulate=False):
    """Winograd form of conv_bwd_weight (3x3 / pad 1): same outputs and workspace handling."""
    _need_dev(x, dy, dw, dbias)
    if dw is None:
        dw = torch.empty(conv_weight_shape(d), dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty((d.n_pad16,), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.efm_wino_wgrad_workspace_bytes(ctypes.byref(d))
    ws = workspace(nbytes, x.device)
    check(lib.efm_wino_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(dbias if want_bias else None), int(bool(accumulate)), _p(ws),
                                  ctypes.c_size_t(ws.numel() * 4), _stream()), "efm_wino_bwd_weight")
    return dw, (dbias if want_bias else None)


def mfmb_fwd(x, c, ways=3):
    """Stand-alone MFM on a bf16 NHWC (pad8) activation."""
    _need_dev(x)
    p8 = lambda v: (v + 7) & ~7  # noqa: E731
    assert x.dtype == torch.bfloat16 and x.shape[-1] == p8(c)
    rows = x.numel() // x.shape[-1]
    out = torch.empty(x.shape[:-1] + (p8(mfm_out_channels(c, ways)),), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().efm_mfmb_fwd(_p(x), _p(out), rows, c, ways, _stream()), "efm_mfmb_fwd")
    return out


def mfmb_bwd(x, dy, c, ways=3, order=_lib.MFM_ORDER_GROUP, add=None):
    _need_dev(x, dy, add)
    rows = x.numel() // x.shape[-1]
    out = torch.empty_like(x)
    check(_lib.load().efm_mfmb_bwd(_p(x), _p(dy), _p(add), _p(out), rows, c, ways, order, _stream()), "efm_mfmb_bwd")
    return out
