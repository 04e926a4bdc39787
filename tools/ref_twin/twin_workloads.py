"""The repo-side tables of the workloads tools/ref_twin/src/main.rs runs on REAL madsim (same names, same programs).

Every main task ends with the fingerprint tail  `trace_instant(); random_u32(); trace_val()`  — the table form of
`fingerprint_tail()` in main.rs: the elapsed time of the main future, then one trailing `random::<u32>()` whose value
depends on every draw before it.  Test infrastructure (used by compare.py and tests/test_ref_twin.py), not product.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from madsim_amd import _abi as A          # noqa: E402
from madsim_amd import workload as W      # noqa: E402

FNV_OFFSET, FNV_PRIME, M64 = 14695981039346656037, 1099511628211, (1 << 64) - 1
VERDICTS = ["pass", "panic", "deadlock", "time-limit", "resource-overflow", "step-limit"]


def fold_obs(values):
    """obs_hash of a madsim_result_t from the list of observed values (MS_OP_TRACE / MS_OP_TRACE_TIME order)."""
    h = FNV_OFFSET
    for v in values:
        h = ((h ^ (v & M64)) * FNV_PRIME) & M64
    return h


def fingerprint_tail(m):
    m.trace_instant()
    m.random_u32()
    m.trace_val()
    m.done()


def pingpong(n_nodes, rounds):
    wl = W.WorkloadBuilder()
    nodes = [wl.create_node() for _ in range(n_nodes)]
    addrs = [wl.addr(n, 1) for n in nodes]
    tasks = []
    for i, n in enumerate(nodes):
        t = wl.task(n)
        t.bind(addrs[i])
        if i % 2 == 0:
            t.sleep(secs=1); t.set(0, rounds); top = t.label()
            t.send_to(addrs[i], addrs[i + 1], 1, W.PING); t.recv_from(addrs[i], 1); t.assert_val(W.PONG); t.djnz(0, top)
        else:
            t.set(0, rounds); top = t.label()
            t.recv_from(addrs[i], 1); t.assert_val(W.PING); t.reply(addrs[i], 1, W.PONG); t.djnz(0, top)
        t.done()
        tasks.append(t)
    m = wl.main()
    for t in tasks:
        m.spawn(t)
    for t in tasks:
        m.join(t)
    fingerprint_tail(m)
    return wl.build()


def sleep_1s():
    wl = W.WorkloadBuilder()
    m = wl.main(); m.sleep(secs=1)
    fingerprint_tail(m)
    return wl.build()


def yield_order():
    wl = W.WorkloadBuilder()
    ts = []
    for i in range(3):
        t = wl.task(0); t.set(0, 5); top = t.label(); t.trace(i * 10, add_reg=0); t.yield_now(); t.djnz(0, top); t.done()
        ts.append(t)
    m = wl.main()
    for t in ts:
        m.spawn(t)
    for t in ts:
        m.join(t)
    fingerprint_tail(m)
    return wl.build()


def timer_ties():
    wl = W.WorkloadBuilder()
    ts = []
    for i in range(6):
        t = wl.task(0)
        for k in range(3):
            t.sleep(ms=10); t.trace(i * 100 + k)
        t.done()
        ts.append(t)
    m = wl.main()
    for t in ts:
        m.spawn(t)
    for t in ts:
        m.join(t)
    fingerprint_tail(m)
    return wl.build()


def _ticker(wl, node, flag, **kw):
    t = wl.task(node, **kw)
    top = t.label(); t.sleep(secs=2); t.flag_add(flag, 2); t.jmp(top)
    return t


def kill():
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    t1, t2 = _ticker(wl, n1, 0), _ticker(wl, n2, 1)
    m = wl.main()
    m.mark(); m.spawn(t1); m.spawn(t2)
    m.sleep_until(secs=3); m.assert_flag(0, 2); m.assert_flag(1, 2)
    m.kill(n1); m.kill(n1); m.assert_exit(n1, True)
    m.sleep_until(secs=5); m.assert_flag(0, 2); m.assert_flag(1, 4)
    fingerprint_tail(m)
    return wl.build()


def restart():
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, init=True)
    t.flag_store(0, 0); top = t.label(); t.sleep(secs=2); t.flag_add(0, 2); t.jmp(top)
    m = wl.main()
    m.mark(); m.build_node(n)
    m.sleep_until(secs=3); m.assert_flag(0, 2)
    m.kill(n); m.restart(n); m.assert_exit(n, False)
    m.sleep_until(secs=6); m.assert_flag(0, 2)
    m.sleep_until(secs=8); m.assert_flag(0, 4)
    fingerprint_tail(m)
    return wl.build()


def restart_on_panic():
    wl = W.WorkloadBuilder()
    n = wl.create_node(restart_on_panic=True)
    t = wl.task(n, init=True)
    t.flag_add(0, 1); t.panic_if_flag_lt(0, 4); t.done()
    m = wl.main()
    m.build_node(n); m.sleep(secs=60); m.assert_flag(0, 4)
    fingerprint_tail(m)
    return wl.build()


def receiver_drop():
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    s = wl.task(n1); s.bind(a1); s.sleep(secs=2); s.send_to(a1, a2, 1, 1); s.done()
    r = wl.task(n2); r.bind(a2); r.recv_from_timeout(a2, 1, secs=1); r.assert_val(A.VAL_TIMEOUT); r.recv_from(a2, 1); r.assert_val(1); r.done()
    m = wl.main()
    m.spawn(s); m.spawn(r); m.join(s); m.join(r)
    fingerprint_tail(m)
    return wl.build()


def localhost():
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    lo1, ip1_2, lo2 = wl.addr(n1, 1, ip="loopback"), wl.addr(n1, 2), wl.addr(n2, 1, ip="loopback")
    ip1_1 = wl.addr(n1, 1)                            # a destination only
    f1 = wl.task(n1); f1.bind(lo1); f1.bind(ip1_2)
    f1.recv_from_timeout(lo1, 1, secs=1); f1.assert_val(A.VAL_TIMEOUT)
    f1.recv_from(ip1_2, 1); f1.assert_val(1); f1.reply(ip1_2, 1, 7); f1.done()
    f2 = wl.task(n2); f2.bind(lo2); f2.sleep(ms=5)
    f2.send_to(lo2, ip1_1, 1, 1); f2.send_to(lo2, ip1_2, 1, 1)
    f2.recv_from_timeout(lo2, 1, secs=2); f2.assert_val(A.VAL_TIMEOUT); f2.done()
    m = wl.main(); m.spawn(f1); m.spawn(f2); m.join(f1); m.join(f2)
    fingerprint_tail(m)
    return wl.build()


def restart_on_panic_matching():
    wl = W.WorkloadBuilder()
    n = wl.create_node(restart_on_panic_matching=(0, 1))
    t = wl.task(n, init=True)
    t.flag_add(0, 1); t.panic_with_flag(0, offset=-1)
    m = wl.main(); m.build_node(n); m.sleep(secs=120)
    fingerprint_tail(m)                                # never reached: the third panic ends the run
    return wl.build()


def bind_ephemeral():
    wl = W.WorkloadBuilder()
    n, other = wl.create_node(), wl.create_node()
    any_a, any_b, any_c = (wl.addr(n, 0, ip="unspecified") for _ in range(3))
    lo_a, foreign, ip100, any3 = wl.addr(n, 0, ip="loopback"), wl.addr(other, 0), wl.addr(n, 100), wl.addr(n, 3, ip="unspecified")
    t = wl.task(n)
    t.bind(any_a, port_to_val=True); t.trace_val()
    t.bind(lo_a, port_to_val=True); t.trace_val()
    t.try_bind(foreign); t.assert_val(A.VAL_ADDR_NOT_AVAILABLE)
    t.bind(ip100, port_to_val=True); t.trace_val(); t.close(ip100); t.bind(ip100)
    t.bind(any_b, port_to_val=True); t.trace_val()
    t.bind(any3); t.bind(any_c, port_to_val=True); t.trace_val()
    t.close(any_a); t.close(any_c)
    t.bind(any_c, port_to_val=True); t.trace_val()
    t.bind(any_a, port_to_val=True); t.trace_val()
    t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    fingerprint_tail(m)
    return wl.build()


def channel_wildcard():
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, dial, acl = wl.addr(ns, 2379, ip="unspecified"), wl.addr(ns, 2379), wl.addr(nc, 0, ip="unspecified")
    srv = wl.task(ns); srv.bind(asv); srv.accept1(asv); srv.set(0, 3)
    top = srv.label(); srv.chan_recv(); srv.trace_val(); srv.chan_send(0x22); srv.djnz(0, top); srv.done()
    cl = wl.task(nc); cl.bind(acl); cl.sleep(ms=10); cl.connect1(acl, dial); cl.assert_val(0); cl.set(0, 3)
    top = cl.label(); cl.chan_send(0x11); cl.chan_recv(); cl.assert_val(0x22); cl.djnz(0, top)
    cl.chan_recv(); cl.assert_val(A.VAL_RESET); cl.done()
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl)
    fingerprint_tail(m)
    return wl.build()


def guard_keeps_address():
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    a7, c1, c2 = wl.addr(ns, 7), wl.addr(nc, 1), wl.addr(nc, 2)
    handler = wl.task(ns); handler.chan_recv(); handler.assert_val(1); handler.sleep(ms=50); handler.chan_send(2); handler.done()
    srv = wl.task(ns); srv.bind(a7); srv.accept1(a7); srv.spawn(handler, move_conn=True)
    srv.close(a7); srv.sleep(ms=5)
    def trace_bind_outcome(t):                    # obs <- 1 if Err(AddrInUse) else 0
        t.try_bind(a7)
        ok = t.label() + 4
        t.jeq(0, ok); t.assert_val(A.VAL_ADDR_IN_USE); t.trace(1); t.jmp(ok + 1)
        assert t.label() == ok
        t.trace(0)
    trace_bind_outcome(srv)
    srv.join(handler); srv.sleep(ms=5)
    trace_bind_outcome(srv)
    srv.done()
    cl = wl.task(nc); cl.bind(c1); cl.sleep(ms=10); cl.connect1(c1, a7); cl.assert_val(0); cl.chan_send(1); cl.sleep(ms=20)
    # the second connection rides on a helper task: one (tx, rx) pair per task in the table form
    second = wl.task(nc); second.bind(c2); second.connect1(c2, a7); second.assert_val(0); second.chan_recv()
    rst = second.label() + 3
    second.jeq(A.VAL_RESET, rst); second.trace(0); second.jmp(rst + 1)
    assert second.label() == rst
    second.trace(1)
    second.done()
    cl.spawn(second); cl.join(second)
    cl.chan_recv(); cl.trace_val(); cl.done()
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(srv); m.join(cl)
    fingerprint_tail(m)
    return wl.build()


def spawn_in_drop_abort():
    """task/mod.rs:1184-1216: the guard of an aborted task spawns a task that runs on the same node (obs <- 1)."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, spawn_on_drop=True)
    c = wl.task(n); c.flag_add(0, 1)
    m = wl.main()
    m.spawn(t); m.abort(t); m.join(t, expect_err=True)
    m.sleep(secs=57257); m.sleep(secs=57257); m.assert_flag(0, 1); m.trace(1)
    fingerprint_tail(m)
    return wl.build()


def spawn_in_drop_kill():
    """task/mod.rs:1219-1253: the guard of a killed node's task spawns a task that never runs (obs <- 1: the guard did drop)."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, spawn_on_drop=True)
    c = wl.task(n); c.panic(7)
    m = wl.main()
    m.spawn(t); m.kill(n); m.join(t, expect_err=True)
    m.sleep(secs=57257); m.sleep(secs=57257); m.trace(1)
    fingerprint_tail(m)
    return wl.build()


def spawn_after_own_restart():
    """Spawner::current() = the caller's own NodeInfo (task/mod.rs:592-599): obs <- child ran 0, saboteur 1, init tasks 2."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    i = wl.task(n, init=True); i.flag_add(2, 1); i.sleep(secs=10)
    c = wl.task(n); c.flag_add(0, 1)
    t = wl.task(n); t.restart(n); t.spawn(c); t.flag_add(1, 1); t.sleep(ms=1); t.flag_add(1, 10)
    m = wl.main(); m.build_node(n); m.sleep(ms=5); m.spawn(t); m.sleep(secs=1)
    m.assert_flag(0, 0); m.trace(0); m.assert_flag(1, 1); m.trace(1); m.assert_flag(2, 2); m.trace(2)
    fingerprint_tail(m)
    return wl.build()


def join_names_its_task():
    """`handle.await` awaits the task the handle named when the await began (task/join.rs:59-72): obs <- 1 worker done."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    worker = wl.task(n); worker.sleep(ms=5); worker.flag_add(0, 1)
    other = wl.task(n); other.sleep(ms=1); other.spawn(worker); other.sleep(ms=1); other.abort(worker); other.sleep(ms=20)
    m = wl.main(); m.mark(); m.spawn(worker); m.spawn(other)
    m.join(worker); m.assert_elapsed(">=", ms=5); m.assert_elapsed("<", ms=7); m.assert_flag(0, 1); m.trace(1)
    m.join(worker, expect_err=True)
    fingerprint_tail(m)
    return wl.build()


def abort_own_handle():
    """A task aborting its own JoinHandle runs on until it yields, then is dropped: obs <- 1."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n); t.sleep(ms=1); t.abort(t); t.flag_add(0, 1); t.sleep(ms=5); t.flag_add(0, 10)
    m = wl.main(); m.spawn(t); m.join(t, expect_err=True); m.sleep(ms=20); m.assert_flag(0, 1); m.trace(1)
    fingerprint_tail(m)
    return wl.build()


def rpc_hooks():
    """NetSim::hook_rpc_req / hook_rpc_rsp (net/mod.rs:240-284; consulted by NetSim::send, :307-311 and :321-328): requests
    Echo(5) leaving node 2 vanish before the link test (no loss / latency draws); every response on its way to node 3 is
    judged by the hook installed when it was SENT and dropped when its timer fires; a replaced hook only judges messages sent
    afterwards; node 4 is untouched.  The handler observes 1 per request it receives."""
    wl = W.WorkloadBuilder()
    ns = wl.create_node(); asv = wl.addr(ns, 1)
    h = wl.task(ns); h.rpc_reply(asv, 42)
    srv = wl.task(ns); srv.bind(asv); top = srv.label()
    srv.rpc_recv(asv, 0); srv.trace(1); srv.spawn(h, move_request=True); srv.jmp(top)
    n1, n2, n3 = wl.create_node(), wl.create_node(), wl.create_node()
    a1, a2, a3 = wl.addr(n1, 1), wl.addr(n2, 1), wl.addr(n3, 1)
    c1 = wl.task(n1); c1.bind(a1); c1.sleep(ms=10)
    c1.rpc_call(a1, asv, 0, 5, timeout_ms=100); c1.assert_val(A.VAL_TIMEOUT)
    c1.rpc_call(a1, asv, 0, 6, timeout_ms=100); c1.assert_val(42)
    c2 = wl.task(n2); c2.bind(a2); c2.sleep(ms=10)
    c2.rpc_call(a2, asv, 0, 7, timeout_ms=100); c2.assert_val(A.VAL_TIMEOUT)
    c2.sleep(ms=400)
    c2.rpc_call(a2, asv, 0, 8, timeout_ms=100); c2.assert_val(42)
    c3 = wl.task(n3); c3.bind(a3); c3.sleep(ms=10); c3.set(0, 3); top = c3.label()
    c3.rpc_call(a3, asv, 0, 9); c3.assert_val(42); c3.djnz(0, top)
    m = wl.main()
    m.hook_rpc_req(n1, 0, code=5); m.hook_rpc_rsp(n2)
    m.spawn(srv); m.spawn(c1); m.spawn(c2); m.spawn(c3)
    m.sleep(ms=300); m.hook_rpc_rsp(n2, code=99)
    m.join(c1); m.join(c2); m.join(c3)
    fingerprint_tail(m)
    return wl.build()


def panic_substrings():
    """`restart_on_panic_matching.iter().any(|s| error_msg.contains(s))` (task/mod.rs:297-300) with literal messages and
    substring patterns: node A ("disk" | "net") and node B ("reset") keep restarting — "network reset" matches both — until
    node C's "out of memory", which its pattern "timeout" does not match, unwinds out of block_on at 100 s."""
    wl = W.WorkloadBuilder()
    a = wl.create_node(restart_on_panic_matching=("disk", "net"))
    b = wl.create_node(restart_on_panic_matching=("reset",))
    c = wl.create_node(restart_on_panic_matching=("timeout",))
    ta2 = wl.task(a); ta2.sleep(secs=7); ta2.panic("network reset")            # task::spawn-ed by A's init task (one init closure per node)
    ta = wl.task(a, init=True); ta.spawn(ta2); ta.flag_add(0, 1); ta.sleep(secs=3); ta.panic("disk full")
    tb = wl.task(b, init=True); tb.flag_add(1, 1); tb.sleep(secs=20); tb.panic("network reset")
    tc = wl.task(c, init=True); tc.sleep(secs=100); tc.panic("out of memory")
    m = wl.main(); m.build_node(a); m.build_node(b); m.build_node(c)
    m.sleep(secs=50); m.panic_if_flag_lt(0, 3); m.panic_if_flag_lt(1, 2); m.trace(1); m.sleep(secs=100)
    fingerprint_tail(m)                                # never reached: "out of memory" ends the run at 100 s
    return wl.build()


def rebind_in_flight():
    """A datagram in flight to a socket that is closed and re-bound before it lands: the delivery closure holds the OLD
    `Arc<dyn Socket>` (net/mod.rs:318-330), so the message lands in the dropped Endpoint's mailbox and the new Endpoint at the
    same address never sees it — its `timeout(recv_from)` elapses (observed value MADSIM_VAL_TIMEOUT) — while a datagram
    sent after the re-bind arrives (observed value 7)."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    rx = wl.task(n2)
    rx.bind(a2); rx.sleep(ms=14)                        # the first datagram is sent at ~12 ms and lands at 13-22 ms;
    rx.close(a2); rx.bind(a2)                           # the drop + re-bind happens at ~15-16 ms: most seeds have it in flight then.
    rx.recv_from_timeout(a2, 1, ms=200); rx.trace_val() # Either way it is lost: landed early = queued in the Endpoint that was dropped
    rx.recv_from_timeout(a2, 2, ms=400); rx.trace_val()
    tx = wl.task(n1)
    tx.bind(a1); tx.sleep(ms=10); tx.send_to(a1, a2, 1, 5)
    tx.sleep(ms=100); tx.send_to(a1, a2, 2, 7)
    m = wl.main(); m.spawn(rx); m.spawn(tx); m.join(rx); m.join(tx)
    fingerprint_tail(m)
    return wl.build()


def ipvs_round_robin():
    """IpVirtualServer (net/ipvs.rs) consulted by NetSim::send and connect1 (net/mod.rs:312-317,345-350): service 1.1.1.1:80 ->
    [10.0.0.1:1, 10.0.0.2:1].  Two connect1("1.1.1.1:80") from node 3 reach node 1 then node 2 (each listener observes what it
    reads), then three datagrams to the same virtual address go 1, 2, 1 — the round-robin index is shared by both paths and
    advances per call (each echo server observes the payloads it sees)."""
    wl = W.WorkloadBuilder()
    n1, n2, n3 = wl.create_node(), wl.create_node(), wl.create_node()
    l1, l2 = wl.addr(n1, 1, ip="unspecified"), wl.addr(n2, 1, ip="unspecified")
    s1, s2 = wl.addr(n1, 1), wl.addr(n2, 1)
    vip = wl.virtual_addr(1, 80)
    wl.ipvs_service(vip, [s1, s2])
    c = wl.addr(n3, 7)
    tasks = []
    for (n, l, base) in ((n1, l1, 100), (n2, l2, 200)):
        t = wl.task(n); t.bind(l); t.accept1(l); t.chan_recv(); t.trace_val()
        t.set(0, 2); top = t.label()
        t.recv_from_timeout(l, 1, ms=300); t.trace_val(); t.djnz(0, top)
        tasks.append(t)
    f3 = wl.task(n3); f3.sleep(ms=50); f3.bind(c)
    f3.connect1(c, vip); f3.assert_val(0); f3.chan_send(1); f3.sleep(ms=30)           # -> node 1
    f3.connect1(c, vip); f3.assert_val(0); f3.chan_send(2); f3.sleep(ms=30)           # -> node 2
    for k in range(3):
        f3.send_to(c, vip, 1, 10 + k); f3.sleep(ms=20)                                # -> node 1, node 2, node 1
    m = wl.main()
    for t in tasks:
        m.spawn(t)
    m.spawn(f3); m.join(tasks[0]); m.join(tasks[1]); m.join(f3)
    fingerprint_tail(m)
    return wl.build()


def ipvs_runtime():
    """IpVirtualServer changed while datagrams flow (net/ipvs.rs:50-105): add_service on a fresh address, add_server x3, three
    sends go a, b, c; del_server(c) and the index left at 3 wraps to a; duplicates ([a, b, a, c]) and retain (del_server(a)
    removes both); del_service stops the rewrite (timeout); add_service on an existing service starts it over.  Every probe =
    send_to(1.1.1.1:80) + timeout(30 ms, recv_from(reply)): the operator observes which server answered (0xA / 0xB / 0xC) or
    0xFFFF_FFFF."""
    wl = W.WorkloadBuilder()
    n1, n2, n3, n4 = (wl.create_node() for _ in range(4))
    vip = wl.virtual_addr(1, 80)
    svc = wl.ipvs_service(vip, absent=True)
    rx, addrs = [], []
    for n, code in ((n1, 0xA), (n2, 0xB), (n3, 0xC)):
        a = wl.addr(n, 1)
        t = wl.task(n); t.bind(a)
        top = t.label(); t.recv_from_timeout(a, 1, ms=400)
        done = t.label() + 3
        t.jeq(A.VAL_TIMEOUT, done); t.reply(a, 2, code); t.jmp(top)
        assert t.label() == done
        rx.append(t); addrs.append(a)
    a, b, c = addrs
    me = wl.addr(n4, 1)
    op = wl.task(n4); op.bind(me); op.sleep(ms=5)

    def probe(times=1):
        for _ in range(times):
            op.send_to(me, vip, 1, 7); op.recv_from_timeout(me, 2, ms=30); op.trace_val()
    op.ipvs_add_service(svc); probe()                              # no servers yet: None
    op.ipvs_add_server(svc, a); op.ipvs_add_server(svc, b); op.ipvs_add_server(svc, c); probe(3)      # a b c
    op.ipvs_del_server(svc, c); probe()                            # [a, b], rr_index 3 -> a
    op.ipvs_add_server(svc, a); op.ipvs_add_server(svc, c); probe(3)                                  # [a, b, a, c] from index 1: b a c
    op.ipvs_del_server(svc, a); probe()                            # [b, c], rr_index 4 -> b
    op.ipvs_del_service(svc); probe()                              # None
    op.ipvs_add_service(svc); probe()                              # fresh, empty
    op.ipvs_add_server(svc, c); probe(2)                           # c c
    m = wl.main()
    for t in rx:
        m.spawn(t)
    m.spawn(op); m.join(op)
    fingerprint_tail(m)
    return wl.build()


def limits(name):
    """Device capacities a table needs beyond the defaults (None = defaults); the oracle has none."""
    if name == "ipvs_runtime":                            # timed-out receives leave dead registrations behind until the next delivery
        lim = A.Limits(); lim.mbox_regs, lim.mbox_msgs = 8, 4
        return lim
    if name == "join_names_its_task":                     # two instances of one program alive at once
        lim = A.Limits(); lim.max_tasks = 6
        return lim
    return None


def ns_ties():
    """Timers that REALLY tie: five pairs of tasks looping `sleep(1 ms + 75 ns)` / `sleep(1 ms)`.  Whenever the two of a pair are
    polled back to back and the 50..100 ns poll cost in between (task/mod.rs:319-321) comes out at 75, their deadlines coincide to
    the nanosecond, and which fires first is the BinaryHeap's array order (naive-timer) — about every seventh seed has
    such a tie.  (`timer_ties` sleeps 10 ms in six tasks whose polls are 50..100 ns apart: close deadlines, no equal ones.)"""
    wl = W.WorkloadBuilder()
    ts = []
    for p in range(5):
        a = wl.task(0); a.set(0, 40); top = a.label(); a.sleep(ms=1, ns=75); a.trace(0x100 + 2 * p); a.djnz(0, top); a.done()
        b = wl.task(0); b.set(0, 40); top = b.label(); b.sleep(ms=1); b.trace(0x101 + 2 * p); b.djnz(0, top); b.done()
        ts += [a, b]
    m = wl.main()
    for t in ts:
        m.spawn(t)
    for t in ts:
        m.join(t)
    fingerprint_tail(m)
    return wl.build()


def update_config_latency():
    """`NetSim::current().update_config(|c| c.send_latency = lo..hi)` between datagrams (net/mod.rs:138-141 -> network.rs:129; every link
    test samples the range in force, :267): 100..101 ms (UniformDuration's Small path), 1.5..3.5 s (Medium path), 1..2 ns.  The receiver
    observes the Instant of every arrival.  Round 6 (the latency half); the ranges: config() below."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a_tx, a_rx = wl.addr(n1, 1), wl.addr(n2, 1)
    rx = wl.task(n2); rx.bind(a_rx)
    for tag in (1, 2, 3, 4):
        rx.recv_from(a_rx, tag); rx.trace_instant()
    rx.done()
    tx = wl.task(n1); tx.mark(); tx.bind(a_tx); tx.sleep(ms=10)
    tx.send_to(a_tx, a_rx, 1, 0xA)
    tx.set_latency(0); tx.send_to(a_tx, a_rx, 2, 0xB)
    tx.set_latency(1); tx.send_to(a_tx, a_rx, 3, 0xC)
    tx.sleep_until(secs=5); tx.set_latency(2); tx.send_to(a_tx, a_rx, 4, 0xD); tx.done()
    m = wl.main(); m.spawn(rx); m.spawn(tx); m.join(rx); m.join(tx)
    fingerprint_tail(m)
    return wl.build()


def self_connect_accept():
    """One task on both ends of its own connection (net/mod.rs:337-364, endpoint.rs:196-212): `ep.connect1(ep's own address)`, a payload
    sent from the client end, then `(tx, rx, _) = ep.accept1()` — the accepted pair replaces the client pair, whose Sender and Receiver
    drop at that assignment: the accepted Receiver yields the payload, then ConnectionReset.  obs <- 5, then 1 for the reset.
    (The round-4 advisor finding — a stale connection header written back in the global-state builds — as a reference twin.)"""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    a = wl.addr(n, 1)
    t = wl.task(n); t.bind(a); t.connect1(a, a); t.assert_val(0); t.chan_send(5); t.sleep(ms=20)
    t.accept1(a); t.chan_recv(); t.trace_val(); t.chan_recv(); t.assert_val(A.VAL_RESET); t.trace(1); t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    fingerprint_tail(m)
    return wl.build()


def config(name, loss=0.0):
    """The Config a twin runs under: Config::default() with the run's packet_loss_rate; `update_config_latency` brings its ranges."""
    if name == "update_config_latency":
        return A.Config.default(packet_loss_rate=loss, lat_table=((100_000_000, 101_000_000), (1_500_000_000, 3_500_000_000), (1, 2)))
    return A.Config.default(packet_loss_rate=loss)


# workloads that end in a panic by design (the reference test is #[should_panic])
EXPECT_PANIC = {"restart_on_panic_matching", "panic_substrings"}

ALL = {
    "pingpong2": lambda: pingpong(2, 64), "pingpong4": lambda: pingpong(4, 64), "pingpong16": lambda: pingpong(16, 8),
    "sleep_1s": sleep_1s, "yield_order": yield_order, "timer_ties": timer_ties, "kill": kill, "restart": restart,
    "restart_on_panic": restart_on_panic, "receiver_drop": receiver_drop,
    "localhost": localhost, "restart_on_panic_matching": restart_on_panic_matching,
    "bind_ephemeral": bind_ephemeral, "channel_wildcard": channel_wildcard, "guard_keeps_address": guard_keeps_address,
    "spawn_in_drop_abort": spawn_in_drop_abort, "spawn_in_drop_kill": spawn_in_drop_kill,
    "spawn_after_own_restart": spawn_after_own_restart,
    "join_names_its_task": join_names_its_task, "abort_own_handle": abort_own_handle,
    # round 3: the semantics added since the first kit, and the table built by the Rust DSL (bindings/rust/madsim-hip)
    "rpc_hooks": rpc_hooks, "panic_substrings": panic_substrings, "rebind_in_flight": rebind_in_flight,
    "ipvs_round_robin": ipvs_round_robin, "ipvs_runtime": ipvs_runtime,
    "pingpong4_dsl": lambda: pingpong(4, 64),          # the same table, built by madsim_hip::pingpong_twin and run by madsim_hip::interp
    "ns_ties": ns_ties,                                # deadlines equal to the nanosecond: the heap's tie order for real
    # round 6: the (f) rows closed since round 3 — run-time send_latency (8f row 1), a task on both ends of its own connection
    "update_config_latency": update_config_latency, "self_connect_accept": self_connect_accept,
}
