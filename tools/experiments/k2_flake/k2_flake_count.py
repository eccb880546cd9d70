import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(3, first_index=1300, h=200, w=280, radius=31, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
def fast():
    _, df = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, return_debug=True)
    return df["rec"].clone(), df["tn"].clone()
os.environ["PVNET_COMPACT_KG"] = "3"; voting.reload_tuning()
ref, tn = fast()
os.environ["PVNET_COMPACT_KG"] = "1"; voting.reload_tuning()
bad = 0
for rep in range(200):
    got, _ = fast()
    bad += not all(torch.equal(ref[bi, :, :int(tn[bi])], got[bi, :, :int(tn[bi])]) for bi in range(3))
print(os.environ.get("PVNET_VOTE_LIB", "default").split("/")[-1], "KG 1 bad", bad, "of 200")
