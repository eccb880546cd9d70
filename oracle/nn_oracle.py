"""CPU oracle of the reference's brute-force nearest-neighbour search  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates lib/utils/extend_utils/src/nearest_neighborhood.cu:48-117 (findNearestPoint3DIdxKernel /
findNearestPoint2DIdxKernel): for every query the reference point of smallest squared distance, the distance evaluated
in float32 in the source's operation order ((x1-x2)^2 + (y1-y2)^2 [+ (z1-z2)^2], one rounding per operation), a strict
`dist < min_dist` scan from index 0 -- i.e. the FIRST minimum wins -- and index 0 when nothing qualifies.
Only tests/ may import this module.  PARITY PINNING: the reference holds no test or golden vector for this function,
so it is pinned by the reference's own device code: nearest_neighborhood.cu is compiled for gfx950 where it lies in the
reference tree (`make -C oracle ref` -> oracle/_ref/libpvnet_refnn.so, through the header shim of oracle/ref_kernels/)
and called through its own launcher `findNearestPointIdxLauncher` on the MI355X; tests/test_evaluation.py holds this
restatement and the product equal to it index for index (ties included), and to scipy's cKDTree in distance."""
import numpy as np


def find_nearest_point_idx(ref_pts, que_pts, exclude_self=False, chunk=512):
    ref = np.ascontiguousarray(ref_pts, np.float32)
    que = np.ascontiguousarray(que_pts, np.float32)
    pn1, pn2 = ref.shape[0], que.shape[0]
    out = np.zeros(pn2, np.int32)
    if pn1 == 0:
        return out
    for q0 in range(0, pn2, chunk):
        q = que[q0:q0 + chunk]
        d = None
        for c in range(ref.shape[1]):  # float32 throughout, the source's left-to-right sum
            diff = ref[None, :, c] - q[:, None, c]
            sq = diff * diff
            d = sq if d is None else d + sq
        if exclude_self:
            rows = np.arange(q.shape[0])
            cols = q0 + rows
            ok = cols < pn1
            d[rows[ok], cols[ok]] = np.inf
        d = np.where(d < np.float32(np.finfo(np.float32).max), d, np.inf)  # `dist < FLT_MAX` never holds for these
        idx = np.argmin(d, axis=1)  # first minimum
        out[q0:q0 + chunk] = np.where(np.isinf(d[np.arange(q.shape[0]), idx]), 0, idx)
    return out
