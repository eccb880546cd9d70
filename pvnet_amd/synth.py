"""Synthetic (mask, unit-vector field) inputs for the voting layer -- SURVEY.md section 8(d).

The field definition follows the reference's ground-truth construction
(lib/datasets/linemod_dataset.py:68-81 ``compute_vertex_hcoords`` / tools/demo.py:58-71 ``compute_vertex``):
for every foreground pixel (x, y) and key-point k the vector (kpt_k - (x, y)) normalised to unit length,
with the ``norm < 1e-3 -> norm += 1e-3`` rule; background vectors are zero ("clean") or N(0,1) ("net-like").
The tensor is stored channel-planar ``[b, 2*vn, h, w]`` -- what the backbone emits
(lib/networks/model_repository.py:76-78) -- and handed to the voting layer as the permuted, NON-contiguous
view ``[b, h, w, vn, 2]`` that tools/demo.py:48-50 builds.

numpy only; used by tests, bench.py and the smoke entry.
"""
from __future__ import annotations

import numpy as np


def disk_mask(h, w, cx, cy, radius):
    ys, xs = np.mgrid[0:h, 0:w]
    return ((xs - cx) ** 2 + (ys - cy) ** 2) <= radius * radius


def field_from_keypoints(fg: np.ndarray, kpts: np.ndarray, background: str = "zeros", rng=None) -> np.ndarray:
    """planar field [2*vn, h, w] float32 for one image (channel 2k = x component, 2k+1 = y component)."""
    h, w = fg.shape
    vn = kpts.shape[0]
    ys, xs = np.nonzero(fg)
    xy = np.stack([xs, ys], axis=1).astype(np.float64)  # (x, y) = (col, row)
    v = kpts[None, :, :2].astype(np.float64) - xy[:, None, :]
    norm = np.linalg.norm(v, axis=2, keepdims=True)
    norm[norm < 1e-3] += 1e-3
    v = v / norm
    if background == "zeros":
        out = np.zeros((h, w, vn, 2), np.float32)
    elif background == "normal":
        out = rng.standard_normal((h, w, vn, 2)).astype(np.float32)
    else:
        raise ValueError(background)
    out[ys, xs] = v.astype(np.float32)
    return np.ascontiguousarray(out.reshape(h, w, vn * 2).transpose(2, 0, 1))


def add_noise(planar: np.ndarray, fg: np.ndarray, rng, sigma_rad=0.05, outlier_frac=0.10):
    """rotate every foreground vector by N(0, sigma) rad and replace a fraction by random unit vectors."""
    c2, h, w = planar.shape
    vn = c2 // 2
    ys, xs = np.nonzero(fg)
    v = planar[:, ys, xs].reshape(vn, 2, -1).astype(np.float64)  # [vn,2,tn]
    ang = rng.normal(0.0, sigma_rad, size=(vn, v.shape[2]))
    ca, sa = np.cos(ang), np.sin(ang)
    vx = ca * v[:, 0] - sa * v[:, 1]
    vy = sa * v[:, 0] + ca * v[:, 1]
    out_sel = rng.random((vn, v.shape[2])) < outlier_frac
    th = rng.uniform(0, 2 * np.pi, size=(vn, v.shape[2]))
    vx = np.where(out_sel, np.cos(th), vx)
    vy = np.where(out_sel, np.sin(th), vy)
    planar = planar.copy()
    planar[:, ys, xs] = np.stack([vx, vy], axis=1).reshape(c2, -1).astype(np.float32)
    return planar


def make_image(index: int, h=480, w=640, vn=9, radius=40, background="zeros", noise=False,
               mask_dtype=np.int64, seed_base=20240, noise_sigma=0.05, outlier_frac=0.10):
    """One synthetic image as SURVEY.md section 8(d) specifies.  Returns (mask [h,w], planar [2vn,h,w], kpts [vn,2])."""
    rng = np.random.default_rng(seed_base + index)
    mx = min(100, w // 4)
    my = min(100, h // 4)
    cx = rng.uniform(mx, w - mx)
    cy = rng.uniform(my, h - my)
    fg = disk_mask(h, w, cx, cy, radius)
    kpts = np.stack([rng.uniform(cx - 1.5 * radius, cx + 1.5 * radius, vn),
                     rng.uniform(cy - 1.5 * radius, cy + 1.5 * radius, vn)], axis=1)
    planar = field_from_keypoints(fg, kpts, background, rng)
    if noise:
        planar = add_noise(planar, fg, rng, sigma_rad=noise_sigma, outlier_frac=outlier_frac)   # (defaults: SURVEY.md 8d's noise variant)
    return fg.astype(mask_dtype), planar, kpts


def make_batch(b: int, first_index=0, **kw):
    """(mask [b,h,w], planar [b,2vn,h,w] float32, kpts [b,vn,2] float64)."""
    ms, ps, ks = zip(*(make_image(first_index + i, **kw) for i in range(b)))
    return np.stack(ms), np.stack(ps), np.stack(ks)


def planar_to_vertex_view(planar):
    """[b,2vn,h,w] -> the non-contiguous [b,h,w,vn,2] view of tools/demo.py:48-50 (numpy array or torch tensor)."""
    b, c2, h, w = planar.shape
    if isinstance(planar, np.ndarray):
        s = planar.strides
        return np.lib.stride_tricks.as_strided(planar, (b, h, w, c2 // 2, 2), (s[0], s[2], s[3], 2 * s[1], s[1]))
    return planar.permute(0, 2, 3, 1).view(b, h, w, c2 // 2, 2)
