"""GPU: the drop-in mode the solvers use -- several processes (PolyChord's MPI ranks, cli.py:110), each with its own
context on the one GPU, each calling lnlhood_pc one theta at a time -- gives every process the same bits a single
process gets, within the parity bar of the oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_two_concurrent_processes_get_the_bits_of_one():
    import dropin_ranks
    one = dropin_ranks.run("B", 1, 150)
    two = dropin_ranks.run("B", 2, 150)
    assert one["bit_equal_across_ranks"] and two["bit_equal_across_ranks"]
    assert one["shared_logL_rank0"] == two["shared_logL_rank0"]           # 32 thetas, bit for bit
    assert max(one["max_abs_dlogL_vs_oracle"], two["max_abs_dlogL_vs_oracle"]) < 1e-4
    assert two["aggregate_logL_per_s"] > 0
