"""CPU oracle for PVNet's RANSAC voting layer  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``pvnet_amd``) never does and fails loudly when its HIP library is missing.

This is a numpy restatement of the reference algorithm (all citations relative to the reference tree):

* ``lib/ransac_voting_gpu_layer/ransac_voting_gpu.py:514-598``  ``ransac_voting_layer_v3`` (driver)
* ``lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu:11-49``   ``generate_hypothesis_kernel``
* ``lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu:88-126``  ``voting_for_hypothesis_kernel``
* ``lib/ransac_voting_gpu_layer/ransac_voting_gpu.py:503-512``       ``b_inv`` (2x2 solve)

PARITY PINNING.  The reference ships no tests and no golden vectors, but both halves of it are executed and
compared:

* its DEVICE CODE -- ransac_voting_kernel.cu is compiled for gfx950 where it lies in the reference tree
  (``make -C oracle ref`` -> oracle/_ref/, header shim in oracle/ref_kernels/) and run on the MI355X:
  tests/test_reference_kernels.py holds the plain-C restatement, this module's float32 flavour, the product's ops
  and the product's literal mode bit-equal to it (hypotheses, inlier flags, counts, winners);
* its PYTHON DRIVER -- ransac_voting_gpu.py is imported from the reference tree and ``ransac_voting_layer_v3``
  executed on CPU tensors (oracle/ref_driver.py; the extension's two kernels stubbed by the C restatement above);
  its outputs, the idxs it drew and the pixels it kept are fixture G6 (tests/golden/ref_driver_v3.npz), which this
  module and the HIP path reproduce within 1e-3 px (tests/test_reference_driver.py).

Further pins (tests/test_oracle.py):

* G1 -- the reference's own demo fixture (data/demo/cat_mask.png + cat_pose.npy + cat_points_3d.txt):
  the ground-truth field built as tools/demo.py:58-71 does must vote back to the analytically projected
  key-points (base_utils.py:252-256) -- this is the known-answer check the reference's ``__main__`` smoke block
  (ransac_voting_gpu.py:1038-1067) prints by eye.
* G4 -- hand-computed op-level cases (two-line intersections incl. the parallel -> (0,0) early-out of
  ransac_voting_kernel.cu:42-43, cosine tests either side of the threshold of :123-124).
* the plain-C restatement in oracle/oracle_c/ written independently from the same reference lines must agree
  bit-for-bit with the float32 path here.

Two arithmetic flavours are offered:

``dtype=np.float64``  "oracle64": every operation in float64 on the float32 inputs -- the semantic anchor.
``dtype=np.float32``  "oracle32": the reference kernels' float32 operation order with one IEEE rounding
                      per multiply/add/sqrt/divide (no FMA contraction -- what ``nvcc -fmad=false`` or numpy
                      float32 gives).  The HIP "literal" mode and the C restatement are bit-exact with this.

The RNG is an *input* here (``idxs`` / ``keep``), exactly as SURVEY.md section 7 (hard part 1) prescribes; when
the caller passes none, the counter-based generator of ``pvnet_amd/csrc/pvnet_rng.h`` is restated below so
that fast-mode device runs are reproducible on the CPU as well.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------------
# Counter-based RNG (restatement of pvnet_amd/csrc/pvnet_rng.h -- the device's curand-free generator)
# ----------------------------------------------------------------------------------------------------
TAG_HYP = 0x48595031  # 'HYP1' : pixel-pair draws, replaces torch's random_() at ransac_voting_gpu.py:547
TAG_SUB = 0x53554231  # 'SUB1' : Bernoulli subsample, replaces uniform_() at ransac_voting_gpu.py:538
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """xorshift-multiply finaliser (32-bit, bijective)."""
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x21F0AAAD)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x735A2D97)) & _M32
    x ^= x >> np.uint64(15)
    return x


def rng_u32(seed: int, tag: int, stream, counter):
    """32 random bits for (seed, tag, stream, counter); all arithmetic mod 2**32."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    stream = np.asarray(stream, dtype=np.uint64)
    counter = np.asarray(counter, dtype=np.uint64)
    x = _mix32((lo ^ np.uint64(tag)) + np.zeros_like(stream + counter))
    x = _mix32(((x ^ ((stream * np.uint64(0x9E3779B1)) & _M32)) + hi) & _M32)
    x = _mix32(x ^ ((counter * np.uint64(0x85EBCA77)) & _M32))
    return x.astype(np.uint32)


def draw_idxs(seed: int, image: int, hn: int, vn: int, tn: int) -> np.ndarray:
    """Pixel-pair indices ``[hn, vn, 2]`` int32 in ``[0, tn)`` (ransac_voting_gpu.py:547, drawn ONCE per image)."""
    cnt = np.arange(hn * vn * 2, dtype=np.uint64)
    r = rng_u32(seed, TAG_HYP, np.uint64(image), cnt).astype(np.uint64)
    return ((r * np.uint64(tn)) >> np.uint64(32)).astype(np.int32).reshape(hn, vn, 2)


THIN_LOG_BINS = 416            # 26 octaves (leading one at bit 0 .. 25) x 16 bins
THIN_LAST = THIN_LOG_BINS - 16 + 1023


def thin_bin(r: int) -> int:
    """bin of a 32-bit random word (pvnet_amd/csrc/pvnet_rng.h: pvnet_thin_bin): steps of 1/1024 of the probability down to
    1/64 (r >> 22 = 16 .. 1023 -> bins 416 .. 1423), sixteen bins per octave of r below; monotone in r."""
    r = int(r)
    if r >> 26:
        return THIN_LOG_BINS - 16 + (r >> 22)
    if r == 0:
        return 0
    e = r.bit_length() - 1
    sub = (r >> (e - 4)) & 15 if e >= 4 else (r << (4 - e)) & 15
    return e * 16 + sub


def subsample_threshold(max_num: int, tn0: int) -> int:
    """keep  <=>  rng_u32 < threshold.  ransac_voting_gpu.py:537-540 keeps a pixel with probability max_num / tn0 when
    tn0 > max_num; here the probability is rounded UP to the next edge of a bin table (thin_bin) -- K = thin_bin(T - 1) + 1
    bins are kept, T = ceil(2^32 max_num / tn0), and the threshold is the first word of bin K -- so that the mask kernel can
    count, per 4096-pixel segment, how many pixels every one of the 1424 possible decisions keeps (a cumulative histogram of
    the bins) before tn0 is known, and compaction needs no separate thinning launch.  Round 4: the table has 1/1024 steps of
    the probability down to 1/64 and relative steps of <= 1/16 below (rounds 2-3: 1/1024 steps only, +17 .. +58 % pixels at
    max_num = 100).  Expected kept pixels: in [max_num, min(max_num + tn0 / 1024, max_num 17 / 16)]."""
    if tn0 <= max_num:
        return 1 << 32
    t = ((int(max_num) << 32) + int(tn0) - 1) // int(tn0)
    k = 0 if t == 0 else thin_bin(t - 1) + 1
    if k > THIN_LAST:
        return 1 << 32
    lo, hi = 0, 1 << 32            # the first word whose bin is >= k (thin_bin is monotone)
    while lo < hi:
        mid = (lo + hi) // 2
        if thin_bin(mid) >= k:
            hi = mid
        else:
            lo = mid + 1
    return lo


def subsample_keep(seed: int, image: int, npix: int, max_num: int, tn0: int) -> np.ndarray:
    """Bernoulli keep flags for every pixel index of an image (ransac_voting_gpu.py:537-540)."""
    thr = subsample_threshold(max_num, tn0)
    if thr >= 2 ** 32:
        return np.ones(npix, bool)
    r = rng_u32(seed, TAG_SUB, np.uint64(image), np.arange(npix, dtype=np.uint64)).astype(np.uint64)
    return r < np.uint64(thr)


# ----------------------------------------------------------------------------------------------------
# Stage restatements
# ----------------------------------------------------------------------------------------------------
def foreground(mask_b: np.ndarray) -> np.ndarray:
    """``mask.byte() != 0`` (ransac_voting_gpu.py:527): integer masks wrap mod 256, floats truncate."""
    m = np.asarray(mask_b)
    if m.dtype == np.bool_:
        return m.copy()
    if np.issubdtype(m.dtype, np.floating):
        return (np.trunc(m).astype(np.int64) & 0xFF) != 0
    return (m.astype(np.int64) & 0xFF) != 0


def compact(fg: np.ndarray, vertex_b: np.ndarray):
    """nonzero / masked_select in raster order (ransac_voting_gpu.py:542-546).

    returns coords [tn,2] float32 as (x=col, y=row) and direct [tn,vn,2] float32."""
    ys, xs = np.nonzero(fg)  # row-major order == torch.nonzero order
    coords = np.stack([xs, ys], axis=1).astype(np.float32)
    direct = np.ascontiguousarray(vertex_b[ys, xs]).astype(np.float32)  # [tn,vn,2]
    return coords, direct


def generate_hypothesis(direct, coords, idxs, dtype=np.float64):
    """ransac_voting_kernel.cu:11-49.  direct [tn,vn,2], coords [tn,2], idxs [hn,vn,2] -> [hn,vn,2].

    Degenerate pairs (either determinant < 1e-6 in magnitude) keep the zero initialisation of :75."""
    T = dtype
    hn, vn, _ = idxs.shape
    k = np.arange(vn)[None, :]
    t0, t1 = idxs[..., 0], idxs[..., 1]
    d0 = direct[t0, k].astype(T)  # [hn,vn,2]
    d1 = direct[t1, k].astype(T)
    c0 = coords[t0].astype(T)
    c1 = coords[t1].astype(T)
    nx0, ny0 = d0[..., 1], -d0[..., 0]  # :31-32
    nx1, ny1 = d1[..., 1], -d1[..., 0]  # :36-37
    cx0, cy0, cx1, cy1 = c0[..., 0], c0[..., 1], c1[..., 0], c1[..., 1]
    det_y = nx1 * ny0 - nx0 * ny1  # :42
    det_x = ny1 * nx0 - ny0 * nx1  # :43
    ok = ~((np.abs(det_y).astype(np.float64) < 1e-6) | (np.abs(det_x).astype(np.float64) < 1e-6))
    b0 = nx0 * cx0 + ny0 * cy0
    b1 = nx1 * cx1 + ny1 * cy1
    with np.errstate(divide="ignore", invalid="ignore"):
        y = (nx1 * b0 - nx0 * b1) / det_y  # :44
        x = (ny1 * b0 - ny0 * b1) / det_x  # :45
    out = np.zeros((hn, vn, 2), T)
    out[..., 0] = np.where(ok, x, 0)
    out[..., 1] = np.where(ok, y, 0)
    return out.astype(T)


def _inlier_block(direct_k, coords, hyp_k, thresh, T):
    """inlier flags [hc, tn] for one key-point: ransac_voting_kernel.cu:107-125."""
    nx = direct_k[None, :, 0].astype(T)
    ny = direct_k[None, :, 1].astype(T)
    dx = hyp_k[:, None, 0].astype(T) - coords[None, :, 0].astype(T)  # :116
    dy = hyp_k[:, None, 1].astype(T) - coords[None, :, 1].astype(T)  # :117
    norm1 = np.sqrt(nx * nx + ny * ny)  # :119
    norm2 = np.sqrt(dx * dx + dy * dy)  # :120
    valid = ~((norm1.astype(np.float64) < 1e-6) | (norm2.astype(np.float64) < 1e-6))  # :121
    with np.errstate(divide="ignore", invalid="ignore"):
        ang = (dx * nx + dy * ny) / (norm1 * norm2)  # :123
    return valid & (ang > T(np.float32(thresh)))  # :124   (thresh arrives as a C float in the reference)


def voting_for_hypothesis(direct, coords, hyp, thresh, dtype=np.float64, hyp_chunk=64):
    """Materialised inlier tensor [hn,vn,tn] uint8 (what the reference op writes; small cases only)."""
    hn, vn, _ = hyp.shape
    tn = coords.shape[0]
    out = np.zeros((hn, vn, tn), np.uint8)
    for k in range(vn):
        for h0 in range(0, hn, hyp_chunk):
            out[h0:h0 + hyp_chunk, k] = _inlier_block(direct[:, k], coords, hyp[h0:h0 + hyp_chunk, k], thresh, dtype)
    return out


def generate_hypothesis_vanishing_point(direct, coords, idxs, dtype=np.float64):
    """ransac_voting_kernel.cu:170-229.  direct [tn,vn,2], coords [tn,2], idxs [hn,vn,2] -> [hn,vn,3]: the homogeneous
    intersection (x, y, z) of the two rays (cross product of their line coordinates), flipped when both rays point away
    from it (:221-222), zeroed when the rays do not meet (:224-225)."""
    T = dtype
    vn = idxs.shape[1]
    k = np.arange(vn)[None, :]
    t0, t1 = idxs[..., 0], idxs[..., 1]
    d0, d1 = direct[t0, k].astype(T), direct[t1, k].astype(T)
    c0, c1 = coords[t0].astype(T), coords[t1].astype(T)
    dx0, dy0, dx1, dy1 = d0[..., 0], d0[..., 1], d1[..., 0], d1[..., 1]
    cx0, cy0, cx1, cy1 = c0[..., 0], c0[..., 1], c1[..., 0], c1[..., 1]
    with np.errstate(over="ignore", invalid="ignore"):
        lx0, ly0, lz0 = dy0, -dx0, cy0 * dx0 - cx0 * dy0  # :202-204
        lx1, ly1, lz1 = dy1, -dx1, cy1 * dx1 - cx1 * dy1  # :206-208
        x = ly0 * lz1 - lz0 * ly1  # :211-213
        y = lz0 * lx1 - lx0 * lz1
        z = lx0 * ly1 - ly0 * lx1
        vx0, vx1 = dx0 * (x - z * cx0), dx1 * (x - z * cx1)  # :216-219
        vy0, vy1 = dy0 * (y - z * cy0), dy1 * (y - z * cy1)
        flip = (vx0 < 0) & (vx1 < 0) & (vy0 < 0) & (vy1 < 0)
        miss = (vx0 * vx1 < 0) | (vy0 * vy1 < 0)
    out = np.stack([x, y, z], axis=-1)
    out = np.where(flip[..., None], -out, out)
    out = np.where(miss[..., None], T(0), out)
    return out.astype(T)


def voting_for_hypothesis_vanishing_point(direct, coords, hyp, thresh, dtype=np.float64, hyp_chunk=64):
    """ransac_voting_kernel.cu:268-310: inlier tensor [hn,vn,tn] uint8 for homogeneous hypotheses hyp [hn,vn,3]."""
    T = dtype
    hn, vn, _ = hyp.shape
    tn = coords.shape[0]
    out = np.zeros((hn, vn, tn), np.uint8)
    cx, cy = coords[None, :, 0].astype(T), coords[None, :, 1].astype(T)
    for k in range(vn):
        ux, uy = direct[None, :, k, 0].astype(T), direct[None, :, k, 1].astype(T)
        norm1 = np.sqrt(ux * ux + uy * uy)
        for h0 in range(0, hn, hyp_chunk):
            hp = hyp[h0:h0 + hyp_chunk, k].astype(T)
            with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
                dx = hp[:, None, 0] - cx * hp[:, None, 2]  # :295-296
                dy = hp[:, None, 1] - cy * hp[:, None, 2]
                norm2 = np.sqrt(dx * dx + dy * dy)
                valid = ~((norm1.astype(np.float64) < 1e-6) | (norm2.astype(np.float64) < 1e-6))  # :300
                ang = (ux * dx + uy * dy) / (norm1 * norm2)
                vx, vy = dx * ux, dy * uy
                wrong = (vx < 0) | (vy < 0)  # :306
                out[h0:h0 + hyp_chunk, k] = valid & ~wrong & (np.abs(ang) > T(np.float32(thresh)))
    return out


def voting_counts(direct, coords, hyp, thresh, dtype=np.float64, hyp_chunk=64):
    """``torch.sum(cur_inlier, 2)`` without materialising it (ransac_voting_gpu.py:557-561) -> [hn,vn] int64."""
    hn, vn, _ = hyp.shape
    counts = np.zeros((hn, vn), np.int64)
    for k in range(vn):
        for h0 in range(0, hn, hyp_chunk):
            counts[h0:h0 + hyp_chunk, k] = _inlier_block(
                direct[:, k], coords, hyp[h0:h0 + hyp_chunk, k], thresh, dtype).sum(axis=1)
    return counts


def refine(direct, coords, win_pts, thresh, dtype=np.float64, lsq_dtype=np.float64):
    """ransac_voting_gpu.py:579-595: inliers of the winners, then per key-point  (sum n n^T) x = sum n (n.c).

    ``lsq_dtype=float64`` accumulates/solves in float64; ``float32`` reproduces the reference's
    un-centred float32 accumulation (its own noise floor, SURVEY.md section 7 hard part 3).
    Returns (pts [vn,2] float32, inlier_counts [vn], singular [vn] bool).  A singular normal matrix makes
    the reference raise inside ``torch.gesv``; here the winning hypothesis is returned and flagged."""
    vn = win_pts.shape[0]
    out = np.zeros((vn, 2), np.float32)
    cnts = np.zeros(vn, np.int64)
    sing = np.zeros(vn, bool)
    L = lsq_dtype
    for k in range(vn):
        inl = _inlier_block(direct[:, k], coords, win_pts[k:k + 1].astype(dtype), thresh, dtype)[0]
        cnts[k] = int(inl.sum())
        n = np.stack([direct[inl, k, 1], -direct[inl, k, 0]], axis=1).astype(L)  # :579-581
        c = coords[inl].astype(L)
        b = (n * c).sum(axis=1, dtype=L)  # :591
        ATA = (n.T @ n).astype(L)  # :592
        ATb = (n * b[:, None]).sum(axis=0, dtype=L)  # :593
        det = ATA[0, 0] * ATA[1, 1] - ATA[0, 1] * ATA[1, 0]
        if cnts[k] == 0 or not np.isfinite(det) or det == 0:
            sing[k] = True
            out[k] = win_pts[k].astype(np.float32)
            continue
        inv = np.array([[ATA[1, 1], -ATA[0, 1]], [-ATA[1, 0], ATA[0, 0]]], L) / det  # b_inv, :503-512
        out[k] = (inv @ ATb).astype(np.float32)  # :594
    return out, cnts, sing


# ----------------------------------------------------------------------------------------------------
# Driver
# ----------------------------------------------------------------------------------------------------
def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                           min_num=5, max_num=30000, *, idxs=None, keep=None, seed=0, image_offset=0, dtype=np.float64,
                           lsq_dtype=np.float64, emulate_rounds=False, return_debug=False):
    """ransac_voting_gpu.py:514-598 on numpy arrays.

    mask [b,h,w] (any dtype), vertex [b,h,w,vn,2] float32 (any strides) -> [b,vn,2] float32.

    ``idxs``  optional [b,hn,vn,2] (or [hn,vn,2]) int pixel-pair indices; default = draw_idxs(seed, image,...).
    ``keep``  optional [b,h,w] bool subsample decisions used when an image has more than ``max_num``
              foreground pixels; default = subsample_keep(seed, image, ...).
    ``emulate_rounds``  run the reference's ``while True`` loop (:552-576) literally; it re-uses the same
              ``idxs`` every round (:547 is outside the loop) so this never changes the result -- kept to prove it.
    """
    mask = np.asarray(mask)
    vertex = np.asarray(vertex)
    b, h, w, vn, _ = vertex.shape
    hn = int(round_hyp_num)
    out = np.zeros((b, vn, 2), np.float32)
    dbg = []
    for bi in range(b):
        fg = foreground(mask[bi])  # :527
        tn0 = int(fg.sum())  # :528
        info = dict(tn0=tn0, skipped=False)
        if tn0 < min_num:  # :531-534
            info["skipped"] = True
            info["tn"] = 0
            dbg.append(info)
            continue
        if tn0 > max_num:  # :537-540
            kp = (np.asarray(keep[bi], bool).reshape(h, w) if keep is not None
                  else subsample_keep(seed, image_offset + bi, h * w, max_num, tn0).reshape(h, w))
            fg = fg & kp
        coords, direct = compact(fg, vertex[bi])  # :542-546
        tn = coords.shape[0]
        info["tn"] = tn
        if tn == 0:  # reference: random_(0,0) raises; defined here as "skipped"
            info["skipped"] = True
            dbg.append(info)
            continue
        if idxs is None:
            ix = draw_idxs(seed, image_offset + bi, hn, vn, tn)  # :547
        else:
            ix = np.asarray(idxs)
            ix = ix[bi] if ix.ndim == 4 else ix
        ix = ix.astype(np.int64)
        all_ratio = np.zeros(vn, np.float32)  # :548
        all_pts = np.zeros((vn, 2), dtype)  # :549
        hyp_num, cur_iter = 0, 0
        while True:  # :552
            hyp = generate_hypothesis(direct, coords, ix, dtype)  # :554
            counts = voting_counts(direct, coords, hyp, inlier_thresh, dtype)  # :557-561
            win_idx = counts.argmax(axis=0)  # :562   (ties: first index -- SURVEY Appendix A.7)
            win_cnt = counts[win_idx, np.arange(vn)]
            win_pts = hyp[win_idx, np.arange(vn)]  # :563
            ratio = win_cnt.astype(np.float32) / np.float32(tn)  # :564
            larger = all_ratio < ratio  # :567
            all_pts[larger] = win_pts[larger]  # :568
            all_ratio[larger] = ratio[larger]  # :569
            hyp_num += hn
            cur_iter += 1
            min_ratio = float(all_ratio.min())
            if not emulate_rounds:
                break  # later rounds regenerate identical hypotheses (idxs fixed) -> no effect on the result
            if (1 - (1 - min_ratio ** 2) ** hyp_num) > confidence or cur_iter > max_iter:  # :575
                break
        pts, inl_cnt, sing = refine(direct, coords, all_pts, inlier_thresh, dtype, lsq_dtype)  # :579-595
        out[bi] = pts
        info.update(hyp=hyp, counts=counts, win_idx=win_idx, win_cnt=win_cnt, win_pts=all_pts.copy(),
                    refine_cnt=inl_cnt, singular=sing, rounds=cur_iter, coords=coords, direct=direct, idxs=ix)
        dbg.append(info)
    return (out, dbg) if return_debug else out


# ----------------------------------------------------------------------------------------------------
# Siblings exported by the same reference module (SURVEY.md section 8f "next" rows)
# ----------------------------------------------------------------------------------------------------
def ransac_motion_voting(mask, vertex):
    """ransac_voting_gpu.py:960-981: masked mean of (vertex + pixel coordinate)."""
    mask = np.asarray(mask)
    vertex = np.asarray(vertex)
    b, h, w, vn, _ = vertex.shape
    out = np.zeros((b, vn, 2), np.float32)
    for bi in range(b):
        fg = foreground(mask[bi])
        if fg.sum() < 1:
            continue
        coords, direct = compact(fg, vertex[bi])
        out[bi] = (direct.astype(np.float64) + coords[:, None, :].astype(np.float64)).mean(axis=0)
    return out


def vote_confidence(mask, vertex, pts, thresh=0.999, dtype=np.float64):
    """ransac_voting_layer_v5's extra output (ransac_voting_gpu.py:848-850): inlier fraction of the refined
    points at 0.999.  (No subsampling branch here: used on small cases.)"""
    mask = np.asarray(mask)
    vertex = np.asarray(vertex)
    b, h, w, vn, _ = vertex.shape
    conf = np.zeros((b, vn), np.float32)
    for bi in range(b):
        coords, direct = compact(foreground(mask[bi]), vertex[bi])
        if coords.shape[0] == 0:
            continue
        c = voting_counts(direct, coords, pts[bi][None].astype(dtype), thresh, dtype)[0]
        conf[bi] = c.astype(np.float32) / np.float32(coords.shape[0])
    return conf


def estimate_voting_distribution_with_mean(mask, vertex, mean, hn, inlier_thresh=0.99, min_num=5, max_num=30000, *,
                                           idxs=None, seed=0, dtype=np.float64):
    """ransac_voting_gpu.py:333-406 with ONE draw of ``hn`` pixel pairs per key-point (the reference draws
    hn/256 rounds of 256 independent pairs).  Returns cov [b,vn,2,2] float64."""
    mask = np.asarray(mask)
    vertex = np.asarray(vertex)
    b, h, w, vn, _ = vertex.shape
    cov = np.zeros((b, vn, 2, 2))
    for bi in range(b):
        fg = foreground(mask[bi])
        tn0 = int(fg.sum())
        if tn0 < min_num:  # :343-349  zero hypotheses with ratio one
            hyp = np.zeros((hn, vn, 2))
            ratio = np.ones((hn, vn))
        else:
            if tn0 > max_num:
                fg = fg & subsample_keep(seed, bi, h * w, max_num, tn0).reshape(h, w)
            coords, direct = compact(fg, vertex[bi])
            tn = coords.shape[0]
            ix = draw_idxs(seed, bi, hn, vn, tn) if idxs is None else np.asarray(idxs)[bi]
            hyp = generate_hypothesis(direct, coords, ix.astype(np.int64), dtype)
            ratio = voting_counts(direct, coords, hyp, inlier_thresh, dtype).astype(np.float32) / np.float32(tn)
        for k in range(vn):
            r = ratio[:, k].astype(np.float64).copy()
            r[ratio[:, k] < np.float32(ratio[:, k].max()) - np.float32(0.1)] = 0.0  # :394-395
            d = hyp[:, k].astype(np.float64) - np.asarray(mean)[bi, k][None]
            cov[bi, k] = (d * r[:, None]).T @ d / (r.sum() + 1e-3)  # :398-401
    return cov
