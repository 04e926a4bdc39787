"""The workload MODEL's ceilings are a layer on top of the oracle, not part of the restatement.

`oracle.run_batch` (model limits ON) is what every parity test holds the device against: a seed that leaves the device runner's workload
model is MADSIM_UNSUPPORTED there, decided at the same instruction as in the kernel.  `oracle.run_batch_pure` runs the very same
restatement with every ceiling OFF — unbounded Vecs and channels as in the reference — and only RECORDS which ceilings a seed met.
These tests hold the two against each other on every fuzz generator: a seed without an event is byte-identical in both runs (so nothing
else in the oracle depends on a ceiling), a seed with one is MADSIM_UNSUPPORTED in the first — the verdict `tests/parity.py` derives
from the event mask instead of taking it from the oracle's word.  Which verdicts are facts about the DEVICE MODEL and not about madsim:
DESIGN.md §2."""
import random

import numpy as np
import pytest

import oracle
from madsim_amd import _abi as A
from tests import fuzz
from tests import lifecycle_workloads as LW


@pytest.mark.parametrize("case", LW.model_ceiling_workloads(), ids=lambda c: c[0])
def test_each_ceiling_is_an_event_of_the_pure_run_and_a_verdict_of_the_model_run(case):
    name, w, lim, bit = case
    on, _ = oracle.run_batch(w, 0, 16, None, lim)
    pure, ev = oracle.run_batch_pure(w, 0, 16, None, lim)
    assert (ev == bit).all(), (name, ev)
    assert (on["verdict"] == A.UNSUPPORTED).all() and not on["steps"].any()
    # the pure run goes on with the reference's unbounded containers and ends in a verdict of the SIMULATION
    assert (pure["verdict"] <= A.TIME_LIMIT).all() and pure["steps"].all() and pure["rng_calls"].all()
    assert (oracle.expected_of_pure(pure, ev) == on).all()


def test_port0_entry_events():
    """A port-0 entry bound again beside its live Endpoint / used after its socket is gone: facts about what the TABLE can say (an entry
    names one Endpoint), recorded by the pure run like the capacity ceilings."""
    for body, bit in ((lambda t, a: (t.bind(a), t.bind(a)), 128), (lambda t, a: (t.bind(a), t.close(a), t.recv_from_timeout(a, 1, ms=2)), 256)):
        wl = fuzz.W.WorkloadBuilder(); n = wl.create_node()
        a = wl.addr(n, 0, ip="unspecified")
        t = wl.task(n); body(t, a); t.done()
        m = wl.main(); m.spawn(t); m.join(t); m.done()
        w = wl.build()
        on, _ = oracle.run_batch(w, 0, 8, None, fuzz.wide_limits(0))
        pure, ev = oracle.run_batch_pure(w, 0, 8, None, fuzz.wide_limits(0))
        assert ((ev & bit) == bit).all() and (on["verdict"] == A.UNSUPPORTED).all() and (oracle.expected_of_pure(pure, ev) == on).all()      # (the pure run's second bind also takes a port beyond the table's candidates: bit 1024)


GENS = [("random_workload", fuzz.generous_limits, {}), ("random_lifecycle_workload", fuzz.generous_limits, {}), ("random_guard_workload", fuzz.generous_limits, {}),
        ("random_rpc_workload", fuzz.generous_limits, {}), ("random_rpc_workload", fuzz.generous_limits, {"hooks": True}),
        ("random_addr_workload", fuzz.generous_limits, {}), ("random_ephemeral_workload", fuzz.generous_limits, {}),
        ("random_channel_workload", fuzz.generous_limits, {}), ("random_supervisor_workload", fuzz.mixed_limits, {}),
        ("random_mixed_workload", fuzz.mixed_limits, {}), ("random_ipvs_workload", fuzz.generous_limits, {}),
        ("random_ipvs_runtime_workload", fuzz.generous_limits, {}), ("random_timeout_workload", fuzz.mailbox_limits, {}),
        ("random_latency_workload", fuzz.mailbox_limits, {}), ("random_reply_without_receive_workload", fuzz.mailbox_limits, {}),
        ("random_unstructured_workload", fuzz.generous_limits, {}), ("random_unstructured_wide_workload", lambda: fuzz.wide_limits(0), {})]

EVENTS_SEEN = {}


@pytest.mark.parametrize("gen,limits,kw", GENS, ids=[g[0] + ("+hooks" if g[2] else "") for g in GENS])
def test_every_generator_both_ways(gen, limits, kw):
    """600 programs x 8 seeds per generator, model limits on and off: byte-identical wherever the seed met no ceiling, MADSIM_UNSUPPORTED
    (every other field 0) in the model run wherever it met one."""
    n_ev = n = 0
    for k in range(600):
        try:
            r = getattr(fuzz, gen)(random.Random(770_000 + k), **kw)
            w, cfg = r[0], r[1]
            on, _ = oracle.run_batch(w, k * 5, 8, cfg, limits())
        except RuntimeError:
            continue                              # (the op-soup generators write programs validate() refuses)
        pure, ev = oracle.run_batch_pure(w, k * 5, 8, cfg, limits())
        want = oracle.expected_of_pure(pure, ev)
        bad = np.nonzero(want != on)[0]
        assert len(bad) == 0, (gen, k, int(bad[0]), on[bad[0]], want[bad[0]], int(ev[bad[0]]))
        inside = ev == 0
        assert (on["verdict"][inside] != A.UNSUPPORTED).all() and (pure[inside] == on[inside]).all()
        n += 8; n_ev += int((~inside).sum())
        for b in np.unique(ev[~inside]):
            EVENTS_SEEN[int(b)] = EVENTS_SEEN.get(int(b), 0) + 1
    assert n >= 3000, n
    print(f"{gen}: {n} seeds both ways, {n_ev} met a ceiling of the workload model")
