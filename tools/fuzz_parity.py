"""Development aid: 150 random configurations, literal HIP path vs the C oracle (winners and counts must be exact)
and fast vs literal (counts within a few votes).   python tools/fuzz_parity.py   (MI355X)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import cref, ransac_voting_oracle as O
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
bad = 0
for case in range(150):
    rng = np.random.default_rng(5000 + case)
    h, w = int(rng.integers(16, 300)), int(rng.integers(16, 400))
    vn = int(rng.integers(1, 14)); hn = int(rng.choice([8, 31, 64, 100, 128, 257, 512, 1000, 1500]))
    b = int(rng.integers(1, 6)); radius = int(rng.integers(3, max(4, min(h, w) // 2)))
    thresh = float(rng.choice([0.5, 0.9, 0.99, 0.999, 0.9999])); max_num = int(rng.choice([30000, 1000, 150, 40]))
    mdt = rng.choice(["int64", "uint8", "int32"])
    mask, planar, _ = synth.make_batch(b, first_index=9000 + 3 * case, h=h, w=w, vn=vn, radius=radius, noise=bool(rng.integers(0, 2)),
                                       background=str(rng.choice(["normal", "zeros"])), mask_dtype=getattr(np, mdt))
    vnp = synth.planar_to_vertex_view(planar)
    m = torch.from_numpy(mask).to(dev); p = torch.from_numpy(planar).to(dev)
    v = synth.planar_to_vertex_view(p) if rng.integers(0, 2) else synth.planar_to_vertex_view(p).contiguous()
    seed = int(rng.integers(0, 2**40))
    out, dbg = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, literal=True, return_debug=True)
    ref, wi, wc = cref.vote_v3(O.foreground(mask), vnp, hn, thresh, max_num=max_num, seed=seed, return_winners=True)
    live = dbg["nchunks"].cpu().numpy() > 0
    ok = np.array_equal(dbg["win"][:, :, 0].cpu().numpy()[live], wi[live]) and np.array_equal(dbg["win"][:, :, 1].cpu().numpy()[live], wc[live])
    fast, df = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, return_debug=True)
    cd = int((df["counts"] - dbg["counts"]).abs().max())
    fin = bool(torch.isfinite(fast).all())
    if not ok or cd > 12 or not fin:  # (fast vs literal counts drift apart as thresh -> 1: float32 cos is flat there)
        bad += 1
        print("MISMATCH case", case, dict(h=h, w=w, vn=vn, hn=hn, b=b, radius=radius, thresh=thresh, max_num=max_num, mdt=mdt), "winners_ok", ok, "max count diff fast-literal", cd, "finite", fin)
print("fuzz done: 150 cases,", bad, "bad")
