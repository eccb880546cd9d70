"""GPU probe of the exact scoring mode: duration of the scoring stage (device clock stamps) in approx / exact mode for
both cell sizes, and how much the rounding band sends to literal re-evaluation.   python tools/exact_probe.py [--quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402


def main():
    quick = "--quick" in sys.argv
    dev = torch.device("cuda:0")
    for radius, thresholds in ((40, (0.99, 0.999, 0.9)), (97, (0.99,))):
        b = 32 if radius == 40 else 8
        mask, planar, _ = synth.make_batch(b, first_index=0, radius=radius, noise=True, background="normal")
        m = torch.from_numpy(mask).to(dev)
        v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
        for thresh in thresholds:
            row = [f"R={radius} b={b} thresh={thresh}"]
            ms = voting.stage_repeat_ms(m, v, 1024, inlier_thresh=thresh, stage="score", repeats=50 if quick else 200,
                                        approx=True)
            row.append(f"approx {ms * 1e3:7.1f} us")
            for fold in (0, 1):
                os.environ["PVNET_EXACT_FOLD"] = str(fold)
                voting.reload_tuning()
                ms = voting.stage_repeat_ms(m, v, 1024, inlier_thresh=thresh, stage="score",
                                            repeats=50 if quick else 200)
                _, dbg = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=1, return_debug=True,
                                                       band_stats=True)
                cells, tests = dbg["band_stats"]
                total = int(dbg["tn"][:b].sum()) * 9 * 1024
                row.append(f"exact(cell={'tile' if fold else 'item'}) {ms * 1e3:7.1f} us  cells {cells} literal tests "
                           f"{tests} = {tests / total:.2e} of {total:.3g}")
            del os.environ["PVNET_EXACT_FOLD"]
            voting.reload_tuning()
            print(" | ".join(row), flush=True)


if __name__ == "__main__":
    main()
