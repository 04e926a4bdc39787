"""GPU parity: the HIP path through the C-ABI vs the CPU oracle, bit-exact on every result field."""
import numpy as np
import pytest

import oracle
from madsim_amd import _abi as A
from madsim_amd import workload as W

pytestmark = pytest.mark.gpu


def _cmp(hip, w, seed0, count, config=None, limits=None):
    got, summ = hip.run_batch(w, seed0, count, config, limits)
    want, osumm = oracle.run_batch(w, seed0, count, config, limits)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"seed {seed0 + bad[0]}: gpu {got[bad[0]]} != oracle {want[bad[0]]}"
    assert summ.n_failed == osumm.n_failed
    assert summ.first_failing_seed == osumm.first_failing_seed
    assert summ.total_steps == osumm.total_steps
    return got, summ


def test_pingpong_2node_one_seed(hip):
    """BASELINE config 0: 2-node ping-pong, 1 seed, plumbing + bit-exact baseline."""
    _cmp(hip, W.pingpong(2, 64), 0, 1)


def test_pingpong_4node_contiguous(hip):
    _cmp(hip, W.pingpong(4, 64), 0, 4096)


def test_pingpong_4node_65536_sampled(hip):
    """BASELINE config 1: 65 536 seeds on one GPU, cross-checked at 256 sampled seeds k*257 mod 65536."""
    w = W.pingpong(4, 64)
    got, summ = hip.run_batch(w, 0, 65536)
    assert summ.n_failed == 0
    for k in range(256):
        s = (k * 257) % 65536
        want, _ = oracle.run_batch(w, s, 1)
        assert got[s] == want[0], f"seed {s}"


def test_pingpong_loss_first_fail(hip):
    """Fault variant: packet loss => deadlock verdicts; first failing seed must match."""
    cfg = A.Config.default(packet_loss_rate=0.01)
    got, summ = _cmp(hip, W.pingpong(4, 64), 0, 2048, cfg)
    assert summ.n_failed > 0
    assert set(np.unique(got["verdict"])) <= {A.PASS, A.DEADLOCK}


def test_trace_log_bytes(hip):
    """The raw determinism log (rand.rs:64-88) of a seed, byte for byte."""
    w = W.pingpong(2, 4)
    for seed in (0, 1, 12345):
        glog, gres = hip.trace_seed(w, seed)
        olog, ores = oracle.trace_seed(w, seed)
        assert glog == olog
        assert gres.astuple() == ores.astuple()
