"""Synthetic inputs generated ON DEVICE (no host pipeline in the timed path).

The generator is the counter-based splitmix64 of oracle/efm_oracle.py restated with torch int64 arithmetic
(wrapping multiply, logical shifts emulated by masking), so a test can check that device data == oracle data.
Images are U[0,1) like `ImageRecordIter(scale=1./255)` output (ref: train_efm.py:179).
"""
import torch

_GOLD = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64
_M1 = -4658895280553007687    # 0xBF58476D1CE4E5B9
_M2 = -7723592293110705685    # 0x94D049BB133111EB


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _to_i64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def uniform01(n, seed, device="cuda", offset=0):
    """n fp32 values in [0,1): bit-identical to oracle.efm_oracle.uniform01."""
    idx = torch.arange(offset, offset + n, dtype=torch.int64, device=device)
    z = (idx + 1) * _GOLD + _to_i64(int(seed))
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    z = z ^ _lsr(z, 31)
    return _lsr(z, 40).to(torch.float32) * (1.0 / (1 << 24))


def images(batch, channels, size, seed, device="cuda"):
    return uniform01(batch * channels * size * size, seed, device).view(batch, channels, size, size)


def parity_labels(batch, rank=0, images_per_identity=4):
    """Reference batch layout: [B/2 anchors ; B/2 positives], labels duplicated (ref: train_efm.py:95-100).
    B/2 anchors over P = B/8 identities x 4 images; 'Celeb1M shard': ids offset by rank*P."""
    h = batch // 2
    p = max(h // images_per_identity, 1)
    ids = torch.arange(h, dtype=torch.int64) % p + rank * p
    return torch.cat([ids, ids])


def negative_indices(labels, seed):
    """The reference's rejection sampling over the anchor half (ref: train_efm.py:234-239) with an explicit,
    reproducible draw stream; vectorised: first draw whose label differs."""
    h = labels.numel() // 2
    lab = labels[:h].cpu()
    if int((lab != lab[0]).sum()) == 0:
        raise ValueError("the batch holds a single identity: the reference's negative pick would never terminate")
    draws = (uniform01(h * 64, seed, device="cpu") * h).to(torch.int64).clamp_(max=h - 1).view(h, 64)
    ok = lab[draws] != lab[:, None]
    first = ok.to(torch.int64).argmax(dim=1)
    if not bool(ok.any(dim=1).all()):
        raise RuntimeError("no negative found in 64 draws")
    return draws[torch.arange(h), first].to(torch.int32)


def identity_faces(ids, channels, size, seed, noise=0.25, grid=7, device="cuda"):
    """Synthetic 'faces' with identity structure (for runs that report a verification accuracy): each identity is a smooth
    random pattern (a `grid` x `grid` random field, bilinearly enlarged, fixed by the identity id) and every image of it adds
    its own U[-noise, noise) pixel noise; values stay in [0, 1].  ids: int tensor (B,); seed varies the per-image noise."""
    import torch.nn.functional as F
    ids = torch.as_tensor(ids, dtype=torch.int64).cpu()
    cells = channels * grid * grid
    base = torch.stack([uniform01(cells, 0x5EED0000 + int(i), device) for i in ids]).view(len(ids), channels, grid, grid)
    faces = F.interpolate(base, size=(size, size), mode="bilinear", align_corners=False)
    eps = (images(len(ids), channels, size, seed, device) * 2 - 1) * noise
    return (faces * (1 - 2 * noise) + noise + eps).clamp_(0.0, 1.0).contiguous()
