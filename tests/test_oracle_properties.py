"""The reference's own (relational) tests for the hot path, restated as workloads and run on the oracle.
Each test cites the Rust test it mirrors (paths under /root/reference/madsim/src/sim/)."""
import numpy as np

import oracle
from madsim_amd import _abi as A
from madsim_amd import workload as W


def test_random_select_from_ready_tasks():
    """task/mod.rs:1018-1041: 3 tasks x 5 sends with yield_now between; 10 seeds => 10 distinct orders."""
    wl = W.WorkloadBuilder()
    ts = []
    for i in range(3):
        t = wl.task(0)
        t.set(0, 5)
        top = t.label()
        t.trace(i * 10, add_reg=0)
        t.yield_now()
        t.djnz(0, top)
        ts.append(t)
    m = wl.main()
    for t in ts:
        m.spawn(t)
    for t in ts:
        m.join(t)
    out, summ = oracle.run_batch(wl.build(), 0, 10)
    assert summ.n_failed == 0
    assert len(set(out["obs_hash"].tolist())) == 10


def test_deterministic_std_instant():
    """time/system_time.rs:140-154: sleep(1 s) elapsed is seed independent (= 1 s + 50 ns)."""
    wl = W.WorkloadBuilder()
    m = wl.main()
    m.mark(); m.sleep(secs=1); m.assert_elapsed(">=", secs=1); m.assert_elapsed("==", secs=1, ns=50)
    out, summ = oracle.run_batch(wl.build(), 0, 64)
    assert summ.n_failed == 0
    # the *clock* differs per seed (random poll costs), the elapsed time seen by the task does not
    assert len(set(out["clock_ns"].tolist())) > 1


def test_time():
    """time/mod.rs:263-294 (without timeout()): 1 ms floor for sleep(0) and sleep_until(now)."""
    wl = W.WorkloadBuilder()
    m = wl.main()
    m.mark(); m.sleep(ns=0); m.assert_elapsed(">=", ms=1)
    m.mark(); m.sleep_until(ns=0); m.assert_elapsed(">=", ms=1)
    m.mark(); m.sleep(secs=1); m.assert_elapsed(">=", secs=1)
    m.sleep_until(secs=2); m.assert_elapsed(">=", secs=2)
    out, summ = oracle.run_batch(wl.build(), 0, 16)
    assert summ.n_failed == 0


def test_advance():
    """time/mod.rs:297-304 test_advance."""
    wl = W.WorkloadBuilder()
    m = wl.main()
    m.mark(); m.advance(secs=1); m.assert_elapsed(">=", secs=1)
    out, summ = oracle.run_batch(wl.build(), 0, 4)
    assert summ.n_failed == 0


def test_spawn_in_block_on():
    """task/mod.rs:850-856."""
    wl = W.WorkloadBuilder()
    t1, t2 = wl.task(0), wl.task(0)
    m = wl.main()
    m.spawn(t1); m.join(t1); m.spawn(t2); m.join(t2)
    out, summ = oracle.run_batch(wl.build(), 0, 4)
    assert summ.n_failed == 0


def test_block_on_pending_is_deadlock():
    """runtime/mod.rs:120-126 doctest: block_on(pending()) panics 'no events, all tasks will block forever'."""
    wl = W.WorkloadBuilder()
    n1 = wl.create_node()
    a1 = wl.addr(n1, 1)
    t = wl.task(n1); t.bind(a1); t.recv_from(a1, 7)            # nobody ever sends
    m = wl.main(); m.spawn(t); m.join(t)
    out, summ = oracle.run_batch(wl.build(), 0, 8)
    assert (out["verdict"] == A.DEADLOCK).all() and summ.first_failing_seed == 0 and summ.n_failed == 8


def test_time_limit():
    """task/mod.rs:253-258 / MADSIM_TEST_TIME_LIMIT."""
    wl = W.WorkloadBuilder()
    m = wl.main(); m.sleep(secs=10)
    lim = A.Limits(); lim.time_limit_ns = 5 * 10**9
    out, _ = oracle.run_batch(wl.build(), 0, 4, None, lim)
    assert (out["verdict"] == A.TIME_LIMIT).all()
    lim.time_limit_ns = 11 * 10**9
    out, _ = oracle.run_batch(wl.build(), 0, 4, None, lim)
    assert (out["verdict"] == A.PASS).all()


def test_send_recv_out_of_order_tags():
    """net/endpoint.rs:372-408 send_recv: tag 2 is received before tag 1 although sent later."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    s = wl.task(n1); s.bind(a1); s.sleep(ms=100); s.send_to(a1, a2, 1, 1); s.sleep(secs=1); s.send_to(a1, a2, 2, 2)
    r = wl.task(n2); r.bind(a2); r.recv_from(a2, 2); r.assert_val(2); r.recv_from(a2, 1); r.assert_val(1)
    m = wl.main(); m.spawn(s); m.spawn(r); m.join(r)
    out, summ = oracle.run_batch(wl.build(), 0, 32)
    assert summ.n_failed == 0 and (out["msg_count"] == 2).all()


def test_send_recv_bytes_and_truncation():
    """net/endpoint.rs:372-408 send_recv with its byte strings: `send_to(addr2, 1, &[1])`, a 16-byte receive buffer, and
    `assert_eq!(len, 1); assert_eq!(buf[0], 1)`; then a long message into a short buffer (len = min(buf.len(), data.len()),
    endpoint.rs:91-92).  Payloads are interned by the builder: the bytes never steer the simulation."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    long_msg = b"a message longer than the receive buffer"
    assert wl.payload(b"ping") == W.PING and wl.payload(long_msg) == wl.payload(bytes(long_msg)) >= W.PAYLOAD_BASE
    s = wl.task(n1); s.bind(a1); s.sleep(ms=100); s.send_to(a1, a2, 1, wl.payload(bytes([1])))
    s.sleep(secs=1); s.send_to(a1, a2, 2, wl.payload(long_msg))
    r = wl.task(n2); r.bind(a2)
    r.recv_from(a2, 2); r.assert_val(wl.payload(long_msg))                 # the raw payload; what a 16-byte buffer keeps of it:
    got, n = wl.received(long_msg, 16)
    assert n == 16 and got == wl.payload(long_msg[:16]) != wl.payload(long_msg)
    r.recv_from(a2, 1); r.assert_val(wl.payload(bytes([1])))
    assert wl.received(bytes([1]), 16) == (wl.payload(bytes([1])), 1)
    m = wl.main(); m.spawn(s); m.spawn(r); m.join(r)
    w = wl.build()
    assert w.payloads == [long_msg, bytes([1]), long_msg[:16]]
    out, summ = oracle.run_batch(w, 0, 32)
    assert summ.n_failed == 0 and (out["msg_count"] == 2).all()


def test_connect_send_recv_pingpong():
    """net/endpoint.rs:549-584 connect_send_recv: the literal ping/pong, one round."""
    out, summ = oracle.run_batch(W.pingpong(2, 1), 0, 32)
    assert summ.n_failed == 0 and (out["msg_count"] == 2).all()


def test_clog_node_blocks_then_unclog_recovers():
    """net/tcp/mod.rs:104 disconnect_and_recovery shape on the datagram API: clogged sends draw nothing
    (network.rs:261-263) and are dropped; after unclog traffic flows."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    s = wl.task(n1); s.bind(a1); s.sleep(ms=10); s.send_to(a1, a2, 1, 5); s.sleep(secs=2); s.send_to(a1, a2, 1, 6)
    r = wl.task(n2); r.bind(a2); r.recv_from(a2, 1); r.assert_val(6)
    m = wl.main(); m.clog_node(n1, "both"); m.spawn(s); m.spawn(r); m.sleep(secs=1); m.unclog_node(n1, "both"); m.join(r)
    out, summ = oracle.run_batch(wl.build(), 0, 32)
    assert summ.n_failed == 0 and (out["msg_count"] == 1).all()


def test_same_seed_same_result_and_seeds_differ():
    """runtime/mod.rs:178-202 check_determinism: a seed re-run reproduces its RNG log byte for byte."""
    w = W.pingpong(4, 8)
    a, _ = oracle.run_batch(w, 100, 50)
    b, _ = oracle.run_batch(w, 100, 50)
    assert (a == b).all()
    assert len(set(a["trace_hash"].tolist())) == 50
    l1, _ = oracle.trace_seed(w, 7)
    l2, _ = oracle.trace_seed(w, 7)
    assert l1 == l2 and len(l1) > 100


def test_pingpong_cost_model():
    """SURVEY §8a cost model: one round trip = 12 executor steps and 2 messages; ~15 ms of sim time."""
    r8, _ = oracle.run_batch(W.pingpong(2, 8), 0, 64)
    r40, _ = oracle.run_batch(W.pingpong(2, 40), 0, 64)
    assert ((r40["steps"].astype(np.int64) - r8["steps"]) == 12 * 32).all()
    assert ((r40["msg_count"] - r8["msg_count"]) == 2 * 32).all()
    per_round_ms = (r40["clock_ns"].astype(np.float64) - r8["clock_ns"]).mean() / 32 / 1e6
    assert 14.0 < per_round_ms < 16.5


def test_loss_produces_deadlocks_and_min_first_fail():
    cfg = A.Config.default(packet_loss_rate=0.01)
    out, summ = oracle.run_batch(W.pingpong(4, 64), 0, 300, cfg)
    fails = np.nonzero(out["verdict"] != A.PASS)[0]
    assert len(fails) > 0 and summ.n_failed == len(fails) and summ.first_failing_seed == fails[0]
    assert set(out["verdict"][fails].tolist()) == {A.DEADLOCK}


def test_deterministic_std_system_time():
    """time/system_time.rs:122-137: 9 runs on seeds 0,0,0,1,1,1,2,2,2 observe 3 distinct SystemTime values (the base time
    is a per-seed draw) — and, deterministic_std_instant (:140-154), one single Instant-based duration."""
    from tests import lifecycle_workloads as LW
    w = LW.std_system_time()
    seen = set()
    for i in range(9):
        out, _ = oracle.run_batch(w, i // 3, 1)
        assert out["verdict"][0] == A.PASS
        seen.add(int(out["obs_hash"][0]))
    assert len(seen) == 3
    out, _ = oracle.run_batch(w, 0, 64)
    assert len(set(out["obs_hash"].tolist())) == 64                    # 64 seeds, 64 base times


def test_getrandom_should_be_deterministic():
    """rand.rs:331-354: ten runs of one seed draw the same getrandom byte; seeds differ."""
    from tests import lifecycle_workloads as LW
    w = LW.getrandom_deterministic()
    runs = {tuple(oracle.run_batch(w, 42, 1)[0][0].tolist()) for _ in range(10)}
    assert len(runs) == 1
    out, _ = oracle.run_batch(w, 0, 256)
    assert len(set(out["obs_hash"].tolist())) == 256


def test_buggify_rate_bounds():
    """buggify.rs:36-60: 1000 draws at 25 % land in 200..300, at 10 % in 50..150 (asserted inside the workload)."""
    from tests import lifecycle_workloads as LW
    out, _ = oracle.run_batch(LW.buggify_rates(), 0, 64, LW.config("buggify_rates"))
    assert (out["verdict"] == A.PASS).all()
    assert (out["rng_calls"] >= 2001).all()
