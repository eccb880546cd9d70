"""Development aid: one fuzz case, the hypotheses where fast and literal counts differ most, against float64 arithmetic."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ransac_voting_oracle as O
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
case = int(sys.argv[1])
rng = np.random.default_rng(5000 + case)
KNOBS = {"PVNET_SCORE_XCD": ["0", "1"], "PVNET_SCORE_ATOMIC": ["0", "1"], "PVNET_SCORE_WGS_PER_CU": ["0", "2", "8"], "PVNET_COMPACT_KG": ["1", "3", "9"]}
for k, vals in KNOBS.items():
    os.environ[k] = str(rng.choice(vals))
voting.reload_tuning()
h, w = int(rng.integers(16, 300)), int(rng.integers(16, 400)); vn = int(rng.integers(1, 14))
hn = int(rng.choice([8, 31, 64, 100, 128, 257, 512, 1000, 1500])); b = int(rng.integers(1, 6))
radius = int(rng.integers(3, max(4, min(h, w) // 2))); thresh = float(rng.choice([0.5, 0.9, 0.99, 0.999, 0.9999]))
max_num = int(rng.choice([30000, 1000, 150, 40])); mdt = rng.choice(["int64", "uint8", "int32"]); scale = float(rng.choice([1.0, 1.0, 2.0 ** -3, 2.0 ** 9]))
mask, planar, _ = synth.make_batch(b, first_index=9000 + 3 * case, h=h, w=w, vn=vn, radius=radius, noise=bool(rng.integers(0, 2)),
                                   background=str(rng.choice(["normal", "zeros"])), mask_dtype=getattr(np, mdt))
planar = (planar * np.float32(scale)).astype(np.float32)
m = torch.from_numpy(mask).to(dev); p = torch.from_numpy(planar).to(dev)
v = synth.planar_to_vertex_view(p) if rng.integers(0, 2) else synth.planar_to_vertex_view(p).contiguous()
seed = int(rng.integers(0, 2 ** 40))
_, dl = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, literal=True, return_debug=True)
cl, hyp, rec, tn = dl["counts"].clone(), dl["hyp"].clone(), dl["rec"].clone(), [int(x) for x in dl["tn"]]
_, df = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, return_debug=True)
cf = df["counts"].clone()
d = (cf - cl).abs()
print("tn", tn, "thresh", thresh, "max diff", int(d.max()), "hypotheses with diff >= 2:", int((d >= 2).sum()), "of", d.numel())
idx = (d >= 2).nonzero().cpu().numpy()
for bi, k, hi in idx[:12]:
    r = rec[bi, k, :tn[bi]].double().cpu().numpy()
    hx, hy = hyp[bi, k, hi].double().cpu().numpy()
    dx, dy = hx - r[:, 0], hy - r[:, 1]
    n1, n2 = np.hypot(r[:, 2], r[:, 3]), np.hypot(dx, dy)
    ok = (n1 >= 1e-6) & (n2 >= 1e-6)
    with np.errstate(invalid="ignore", divide="ignore"):
        c64 = int((ok & ((dx * r[:, 2] + dy * r[:, 3]) / (n1 * n2) > thresh)).sum())
    print(f"image {bi} kp {k} hyp {hi}: float64 {c64}  fast {int(cf[bi, k, hi])} (err {int(cf[bi, k, hi]) - c64:+d})  literal {int(cl[bi, k, hi])} (err {int(cl[bi, k, hi]) - c64:+d})")
