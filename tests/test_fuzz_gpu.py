"""200 fuzz cases in the driver's GPU suite: the exact (default) mode must return literal mode's integers on
random shapes with the launch knobs flipped at random -- until round 4 only the hand-run tools/fuzz_parity.py asserted that for
odd vn / hn (profiles/r0*_fuzz_*.txt)."""
import pytest

pytestmark = pytest.mark.gpu

FIRST, BLOCK, BLOCKS = 20000, 25, 8   # cases 20000 .. 20199: seeds no earlier round's hand-run fuzz has seen


@pytest.mark.parametrize("block", range(BLOCKS))
def test_fuzz_block_exact_equals_literal_equals_c_oracle(block):
    from tests import fuzz_cases as F
    bad = []
    try:
        for case in range(FIRST + block * BLOCK, FIRST + (block + 1) * BLOCK):
            r = F.run_case(case)
            if not (r["ok_literal"] and r["ok_exact"] and r["approx_diff"] <= r["approx_limit"] and r["finite"]):
                bad.append(r)
    finally:
        F.clear_knobs()
    assert not bad, bad
