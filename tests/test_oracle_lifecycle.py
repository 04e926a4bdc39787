"""Node lifecycle on the oracle: the reference's own unit tests restated."""
import numpy as np
import pytest

import oracle
from madsim_amd import _abi as A
from tests import lifecycle_workloads as LW


@pytest.mark.parametrize("name", sorted(LW.ALL))
def test_reference_lifecycle_test(name):
    out, summ = oracle.run_batch(LW.ALL[name](), 0, 64, LW.config(name))
    want = A.PANIC if name in LW.EXPECT_PANIC else A.PASS
    assert (out["verdict"] == want).all(), (name, np.bincount(out["verdict"]))


def test_restart_on_panic_delays_are_random_per_seed():
    """The 1..10 s restart delay is a GlobalRng draw (task/mod.rs:302-304): seeds must differ."""
    out, _ = oracle.run_batch(LW.restart_on_panic(), 0, 64)
    assert len(set(out["trace_hash"].tolist())) == 64
