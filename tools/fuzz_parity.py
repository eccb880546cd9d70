"""Development aid: thousands of random configurations with the launch knobs flipped at random (the cases live in
tests/fuzz_cases.py; tests/test_fuzz_gpu.py runs 200 of them in the GPU suite) --
  literal HIP path vs the C oracle: winners and their counts must be exact;
  EXACT mode (the default) vs literal: every hypothesis, every count, every winner must be EQUAL, key-points within 1e-3 px;
  approximate mode vs literal: counts within 2 votes up to thresh 0.999.
    python tools/fuzz_parity.py [cases [first_case]]   (MI355X)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import fuzz_cases as F  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
START = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # first case (cases are seeded by their number)
bad_exact = bad = 0
worst = {}
for case in range(START, N):
    r = F.run_case(case)
    worst[r["thresh"]] = max(worst.get(r["thresh"], 0), r["approx_diff"])
    if not r["ok_exact"]:
        bad_exact += 1
        print("EXACT-MODE MISMATCH", r["desc"], flush=True)
    if not r["ok_literal"] or r["approx_diff"] > r["approx_limit"] or not r["finite"]:
        bad += 1
        print("MISMATCH", r["desc"], "winners_ok", r["ok_literal"], "max count diff approx-literal", r["approx_diff"], "finite",
              r["finite"], flush=True)
    if (case + 1) % 100 == 0:
        print(f"... case {case + 1}: exact != literal {bad_exact}, other failures {bad}", flush=True)
F.clear_knobs()
print(f"fuzz done: cases {START}..{N - 1}; exact mode != literal in {bad_exact} cases; literal-vs-C-oracle / approx-bound failures {bad}; "
      f"worst approx-vs-literal count difference per threshold: {dict(sorted(worst.items()))}")
