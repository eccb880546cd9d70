"""Development aid (GPU box): what the reference's own kernels (oracle/_ref, built by `make -C oracle ref`) say on
MI355X -- exactness against our literal mode, sensitivity of the reference itself to FMA contraction, and the time
its two kernels take per image at the BASELINE shape.   python tools/reference_kernels_report.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ransac_voting_oracle as O  # noqa: E402
from oracle import refkernels  # noqa: E402
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
print(refkernels.lib("off").ref_build_info().decode(), "|", refkernels.lib("fast").ref_build_info().decode())
B, HN, TH = 4, 1024, 0.99
mask, planar, _ = synth.make_batch(B, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
out, dbg = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=TH, seed=1, literal=True, return_debug=True)
lit_counts, lit_hyp, lit_win = dbg["counts"].clone(), dbg["hyp"].clone(), dbg["win"].clone()
tns = [int(t) for t in dbg["tn"][:B]]
recs = dbg["rec"].clone()
_, fdbg = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=TH, seed=1, return_debug=True)
fast_counts, fast_win = fdbg["counts"].clone(), fdbg["win"].clone()

tot = dict(pairs=0, hyp_exact=0, cnt_exact=0, flips_fma=0, fast_diff=0, win_lit=0, win_fast=0, kp=0, hyp_n=0)
hyp_rel = 0.0
t_ref = []
for bi in range(B):
    tn = tns[bi]
    rec = recs[bi, :, :tn]
    coords = rec[0, :, 0:2].contiguous()
    direct = rec[:, :, 2:4].permute(1, 0, 2).contiguous()
    idxs = torch.from_numpy(O.draw_idxs(1, bi, HN, 9, tn)).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hyp = refkernels.generate_hypothesis(direct, coords, idxs)
    inl = refkernels.voting_for_hypothesis(direct, coords, hyp, TH)
    counts = inl.sum(2, dtype=torch.int32)
    torch.cuda.synchronize()
    t_ref.append(time.perf_counter() - t0)
    tot["hyp_n"] += hyp.numel()
    tot["hyp_exact"] += int((hyp.view(torch.int32) == lit_hyp[bi].permute(1, 0, 2).contiguous().view(torch.int32)).sum())
    tot["cnt_exact"] += int((counts == lit_counts[bi].T).sum())
    tot["pairs"] += HN * 9 * tn
    hyp_f = refkernels.generate_hypothesis(direct, coords, idxs, contract="fast")
    hyp_rel = max(hyp_rel, float(((hyp - hyp_f).abs() / hyp.abs().clamp_min(1.0)).max()))
    inl_f = refkernels.voting_for_hypothesis(direct, coords, hyp, TH, contract="fast")
    tot["flips_fma"] += int((inl != inl_f).sum())
    tot["fast_diff"] += int((fast_counts[bi].T - counts).abs().sum())
    # float64 truth of the same cosine test on the same float32 inputs (ransac_voting_kernel.cu:113-124 in exact arithmetic)
    c64, u64, h64 = coords.double(), direct.double(), hyp.double()
    cnt64 = torch.zeros_like(counts)
    for k in range(9):
        dxy = h64[:, k, None, :] - c64[None, :, :]                      # [hn,tn,2]
        n1 = u64[:, k].norm(dim=1)[None, :]
        n2 = dxy.norm(dim=2)
        cosv = (dxy * u64[None, :, k, :]).sum(2) / (n1 * n2)
        ok = (n1 >= 1e-6) & (n2 >= 1e-6) & (cosv > float(np.float32(TH)))
        cnt64[:, k] = ok.sum(1).int()
    tot["ref_vs_64"] = tot.get("ref_vs_64", 0) + int((counts - cnt64).abs().sum())
    tot["fast_vs_64"] = tot.get("fast_vs_64", 0) + int((fast_counts[bi].T - cnt64).abs().sum())
    first = (counts == counts.max(0).values[None]).int().argmax(0)
    tot["win_lit"] += int((lit_win[bi, :, 0].long() == first).sum())
    tot["win_fast"] += int((fast_win[bi, :, 0].long() == first).sum())
    tot["kp"] += 9
print(f"images {B}, tn {tns}, hypotheses {HN}, thresh {TH}")
print(f"hypotheses bit-equal to the reference kernel (literal mode): {tot['hyp_exact']}/{tot['hyp_n']}")
print(f"inlier counts equal to the reference kernel (literal mode):  {tot['cnt_exact']}/{HN * 9 * B}")
print(f"winners equal to the reference kernel: literal {tot['win_lit']}/{tot['kp']}, fast {tot['win_fast']}/{tot['kp']}")
print(f"fast mode: sum |count - reference count| = {tot['fast_diff']} over {tot['pairs']} pair tests "
      f"({tot['fast_diff'] / tot['pairs']:.2e})")
print(f"against float64 arithmetic on the same inputs: sum |count diff| reference kernels {tot['ref_vs_64']} "
      f"({tot['ref_vs_64'] / tot['pairs']:.2e}), fast mode {tot['fast_vs_64']} ({tot['fast_vs_64'] / tot['pairs']:.2e})")
print(f"reference, fp-contract fast vs off: max relative hypothesis change {hyp_rel:.2e}; "
      f"inlier flags flipped {tot['flips_fma']} of {tot['pairs']} ({tot['flips_fma'] / tot['pairs']:.2e})")
t = sorted(t_ref)[len(t_ref) // 2]
print(f"reference kernels on MI355X (recompiled, + torch.sum of the [hn,vn,tn] uint8 tensor): {t * 1e3:.2f} ms per image "
      f"-> {1 / t:.0f} votings/s for the two kernels alone (no compaction, arg-max, refinement, host syncs)")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=TH, seed=1)
torch.cuda.synchronize()
print(f"this library, whole path, batch {B}: {(time.perf_counter() - t0) / 20 / B * 1e3:.3f} ms per image")
