"""The reference's node-lifecycle unit tests (madsim/src/sim/task/mod.rs:859-1182) restated as workloads.
Shared by the oracle tests (CPU) and the GPU parity tests."""
from madsim_amd import _abi as A
from madsim_amd import workload as W


def _ticker(wl, node, flag, **kw):
    """`loop { sleep(2 s).await; flag.fetch_add(2) }`"""
    t = wl.task(node, **kw)
    top = t.label()
    t.sleep(secs=2); t.flag_add(flag, 2); t.jmp(top)
    return t


def kill():
    """task/mod.rs:859-897"""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    _ticker(wl, n1, 0, pre=True); _ticker(wl, n2, 1, pre=True)
    m = wl.main()
    m.mark(); m.sleep_until(secs=3); m.assert_flag(0, 2); m.assert_flag(1, 2)
    m.kill(n1); m.kill(n1); m.assert_exit(n1, True)
    m.sleep_until(secs=5); m.assert_flag(0, 2); m.assert_flag(1, 4)
    return wl.build()


def restart():
    """task/mod.rs:899-936"""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, init=True, pre=True)
    t.flag_store(0, 0)
    top = t.label()
    t.sleep(secs=2); t.flag_add(0, 2); t.jmp(top)
    m = wl.main()
    m.mark(); m.sleep_until(secs=3); m.assert_flag(0, 2)
    m.kill(n); m.restart(n); m.assert_exit(n, False)
    m.sleep_until(secs=6); m.assert_flag(0, 2)
    m.sleep_until(secs=8); m.assert_flag(0, 4)
    return wl.build()


def restart_on_panic():
    """task/mod.rs:938-962: panics 3 times, succeeds once; restart delays are random 1..10 s."""
    wl = W.WorkloadBuilder()
    n = wl.create_node(restart_on_panic=True)
    t = wl.task(n, init=True, pre=True)
    t.flag_add(0, 1); t.panic_if_flag_lt(0, 4)
    m = wl.main()
    m.sleep(secs=60); m.assert_flag(0, 4)
    return wl.build()


def restart_on_panic_matching():
    """task/mod.rs:964-982 (#[should_panic(expected = "2")]): the init task panics with its attempt number; the node
    restarts on messages "0" and "1", so the third panic ("2") unwinds out of block_on."""
    wl = W.WorkloadBuilder()
    n = wl.create_node(restart_on_panic_matching=(0, 1))
    t = wl.task(n, init=True, pre=True)
    t.flag_add(0, 1); t.panic_with_flag(0, offset=-1)          # panic!("{}", flag.fetch_add(1))
    m = wl.main(); m.sleep(secs=120)                          # block_on(pending()): the panic ends the run long before
    return wl.build()


def restart_on_panic_matching_other_message():
    """A message no pattern names unwinds at once (no restart, no 1..10 s draw): patterns "5" / "6", panic "0"."""
    wl = W.WorkloadBuilder()
    n = wl.create_node(restart_on_panic_matching=(5, 6))
    t = wl.task(n, init=True, pre=True)
    t.sleep(ms=3); t.panic(code=0)
    m = wl.main(); m.sleep(secs=120)
    return wl.build()


def restart_on_panic_matching_substrings():
    """`restart_on_panic_matching.iter().any(|s| error_msg.contains(s))` (task/mod.rs:297-300) with literal messages and
    substring patterns: nodes A ("disk" | "net") and B ("reset") keep restarting on the messages that contain one of their
    patterns — "network reset" matches both — until node C's "out of memory", which its pattern "timeout" does not match,
    unwinds out of block_on at 100 s."""
    wl = W.WorkloadBuilder()
    a = wl.create_node(restart_on_panic_matching=("disk", "net"))
    b = wl.create_node(restart_on_panic_matching=("reset",))
    c = wl.create_node(restart_on_panic_matching=("timeout",))
    ta = wl.task(a, init=True, pre=True); ta.flag_add(0, 1); ta.sleep(secs=3); ta.panic("disk full")
    ta2 = wl.task(a, init=True, pre=True); ta2.sleep(secs=7); ta2.panic("network reset")
    tb = wl.task(b, init=True, pre=True); tb.flag_add(1, 1); tb.sleep(secs=20); tb.panic("network reset")
    tc = wl.task(c, init=True, pre=True); tc.sleep(secs=100); tc.panic("out of memory")
    m = wl.main(); m.sleep(secs=50); m.panic_if_flag_lt(0, 3); m.panic_if_flag_lt(1, 2); m.sleep(secs=100)
    built = wl.build()
    assert built.struct.panic_dyn_max == 254 - 3 and built.panic_match is not None     # three literal messages, one row per node
    return built


def panic_without_restart():
    """task/mod.rs:315 resume_unwind: a panic on a node that does not restart fails the run."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, init=True, pre=True)
    t.sleep(ms=5); t.panic()
    m = wl.main(); m.sleep(secs=1)
    return wl.build()


def pause_resume():
    """task/mod.rs:985-1015"""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    _ticker(wl, n, 0, pre=True)
    m = wl.main()
    m.mark(); m.sleep_until(secs=3); m.assert_flag(0, 2)
    m.pause(n); m.pause(n)
    m.sleep_until(secs=5); m.assert_flag(0, 2)
    m.resume(n); m.resume(n)
    m.sleep_until(secs=5, ms=500); m.assert_flag(0, 4)
    return wl.build()


def kill_drop_futures():
    """task/mod.rs:1112-1135 (the Arc strong counts are not observable; the scheduling is)."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    a = wl.addr(n, 1)
    t = wl.task(n, pre=True); t.bind(a); t.recv_from(a, 9)          # pending forever
    m = wl.main(); m.sleep(secs=1); m.kill(n); m.sleep(secs=1)
    return wl.build()


def join_cancelled():
    """task/mod.rs:1137-1151"""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    a = wl.addr(n, 1)
    t = wl.task(n, pre=True); t.bind(a); t.recv_from(a, 9)
    m = wl.main(); m.abort(t); m.join(t, expect_err=True)
    return wl.build()


def exited():
    """task/mod.rs:1153-1182: the init future returns => node exits, its other tasks are killed."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    sub = _ticker(wl, n, 0)
    t = wl.task(n, init=True, pre=True)
    t.spawn(sub); t.sleep(secs=5)
    m = wl.main()
    m.assert_exit(n, False); m.sleep(secs=10); m.assert_exit(n, True); m.assert_flag(0, 4)
    return wl.build()


def spawn_on_killed_node():
    """task/mod.rs:1219-1253 shape: NodeHandle::spawn on a killed node succeeds but the task never runs."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n); t.flag_add(0, 1)
    m = wl.main(); m.kill(n); m.spawn(t); m.join(t, expect_err=True); m.sleep(secs=1); m.assert_flag(0, 0)
    return wl.build()


def kill_restart_with_traffic():
    """A server node is killed and restarted by the supervisor while a client keeps pinging it
    (the kill/restart fault loop of tonic-example/tests/test.rs:234-271 on the datagram API)."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    srv = wl.task(ns, init=True, pre=True)
    srv.bind(asv)
    top = srv.label()
    srv.recv_from(asv, 1); srv.flag_add(0, 1); srv.jmp(top)
    cl = wl.task(nc, pre=True)
    cl.bind(acl); cl.set(0, 40)
    top = cl.label()
    cl.send_to(acl, asv, 1, 3); cl.sleep(ms=50); cl.djnz(0, top)
    m = wl.main()
    m.sleep(ms=300); m.kill(ns); m.sleep(ms=200); m.restart(ns); m.sleep(ms=400); m.kill(ns); m.restart(ns)
    m.join(cl)
    return wl.build()


def receiver_drop():
    """net/endpoint.rs:410-444 receiver_drop (Barrier replaced by sleeps): a recv_from wrapped in timeout() elapses,
    its receiver is dropped, and a later recv_from on the same endpoint still gets the message."""
    from madsim_amd import _abi as A
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    s = wl.task(n1); s.bind(a1); s.sleep(secs=2); s.send_to(a1, a2, 1, 1)
    r = wl.task(n2); r.bind(a2); r.recv_from_timeout(a2, 1, secs=1); r.assert_val(A.VAL_TIMEOUT); r.recv_from(a2, 1); r.assert_val(1)
    m = wl.main(); m.spawn(s); m.spawn(r); m.join(r)
    return wl.build()


def request_timeout_with_stale_timers():
    """tonic-example/tests/test.rs:369 request_timeout shape: a reply arrives before the timeout; the timeout's Sleep
    has registered duplicate timers (time/sleep.rs:51-53) that later wake the task spuriously while it sleeps."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    srv = wl.task(n1); srv.bind(a1); srv.recv_from(a1, 1); srv.sleep(ms=300); srv.reply(a1, 2, 9)
    cl = wl.task(n2); cl.bind(a2); cl.sleep(ms=50); cl.send_to(a2, a1, 1, 4); cl.recv_from_timeout(a2, 2, secs=2); cl.assert_val(9)
    cl.mark(); cl.sleep(secs=5); cl.assert_elapsed(">=", secs=5)
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl)
    return wl.build()


def random_restart_loop():
    """tonic-example/tests/test.rs:198-201: `for _ in 0..10 { sleep(gen_range(0..5 s)); handle.restart(node) }`."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    a = wl.addr(n, 1)
    srv = wl.task(n, init=True, pre=True)
    srv.bind(a); srv.flag_add(0, 1)
    top = srv.label()
    srv.sleep(ms=700); srv.flag_add(1, 1); srv.jmp(top)
    m = wl.main()
    m.set(0, 10)
    top = m.label()
    m.sleep_rand(lo_ms=0, secs=5); m.kill(n); m.restart(n); m.djnz(0, top)
    m.sleep(ms=10); m.assert_exit(n, False)
    return wl.build()


ALL = dict(receiver_drop=receiver_drop, request_timeout_with_stale_timers=request_timeout_with_stale_timers,
           random_restart_loop=random_restart_loop,
           kill=kill, restart=restart, restart_on_panic=restart_on_panic, panic_without_restart=panic_without_restart,
           pause_resume=pause_resume, kill_drop_futures=kill_drop_futures, join_cancelled=join_cancelled,
           exited=exited, spawn_on_killed_node=spawn_on_killed_node, kill_restart_with_traffic=kill_restart_with_traffic)
EXPECT_PANIC = {"panic_without_restart", "restart_on_panic_matching", "restart_on_panic_matching_other_message",
                "restart_on_panic_matching_substrings"}


def kv_rpc(n_clients=2, n_ops=3):
    """The request-per-connection shape every higher simulator uses (madsim-etcd-client/src/kv.rs:37-53,
    server.rs:34-40; madsim-tonic client.rs:66): client connect1 -> send request -> recv response;
    server: loop { accept1; spawn(handler with the moved (tx, rx)) }."""
    wl = W.WorkloadBuilder()
    ns = wl.create_node()
    asv = wl.addr(ns, 2379)
    handler = wl.task(ns)
    handler.chan_recv(); handler.assert_val(0x11); handler.flag_add(0, 1); handler.chan_send(0x22)
    srv = wl.task(ns)
    srv.bind(asv)
    top = srv.label()
    srv.accept1(asv); srv.spawn(handler, move_conn=True); srv.jmp(top)
    clients = []
    for i in range(n_clients):
        nc = wl.create_node()
        acl = wl.addr(nc, 1)
        c = wl.task(nc)
        c.bind(acl); c.sleep(ms=10); c.set(0, n_ops)
        top = c.label()
        c.connect1(acl, asv); c.assert_val(0); c.chan_send(0x11); c.chan_recv(); c.assert_val(0x22); c.chan_close(); c.djnz(0, top)
        clients.append(c)
    m = wl.main()
    m.spawn(srv)
    for c in clients:
        m.spawn(c)
    for c in clients:
        m.join(c)
    m.assert_flag(0, n_clients * n_ops)
    return wl.build()


def channel_backoff():
    """net/mod.rs:388-398: a payload sent while the link is clogged is stamped None; the receiver retries with
    1 ms -> 2 -> 4 ... backoff and delivers once the supervisor unclogs the link."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    srv = wl.task(ns); srv.bind(asv); srv.accept1(asv); srv.mark(); srv.chan_recv(); srv.assert_val(7); srv.assert_elapsed(">=", secs=2)
    cl = wl.task(nc); cl.bind(acl); cl.sleep(ms=10); cl.connect1(acl, asv); cl.assert_val(0); cl.sleep(ms=100); cl.chan_send(7); cl.sleep(secs=10)
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.sleep(ms=50); m.clog_link(nc, ns); m.sleep(secs=3); m.unclog_link(nc, ns); m.join(srv)
    return wl.build()


def connect_refused_and_reset():
    """net/mod.rs:351-354 ConnectionRefused (nobody listening); endpoint.rs:243-244 / 259-260 ConnectionReset
    (peer dropped its handles)."""
    from madsim_amd import _abi as A
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    srv = wl.task(ns); srv.sleep(ms=100); srv.bind(asv); srv.accept1(asv); srv.chan_recv(); srv.assert_val(1)     # then drops (tx, rx)
    cl = wl.task(nc); cl.bind(acl); cl.connect1(acl, asv); cl.assert_val(A.VAL_REFUSED)
    cl.sleep(ms=300); cl.connect1(acl, asv); cl.assert_val(0); cl.chan_send(1); cl.chan_recv(); cl.assert_val(A.VAL_RESET)
    cl.sleep(ms=100); cl.chan_send(2); cl.assert_val(A.VAL_RESET)
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl)
    return wl.build()


ALL.update(restart_on_panic_matching_substrings=restart_on_panic_matching_substrings)
ALL.update(restart_on_panic_matching=restart_on_panic_matching,
           restart_on_panic_matching_other_message=restart_on_panic_matching_other_message)
ALL.update(kv_rpc=kv_rpc, channel_backoff=channel_backoff, connect_refused_and_reset=connect_refused_and_reset)


def _rpc_server(wl, node, addr, reply_code, handler_sleep_ms=0, **kw):
    """add_rpc_handler (net/rpc.rs:152-179): `loop { (req, from) = recv_from_raw(R::ID); spawn(async move { rsp = f(req).await;
    send_to_raw(from, rsp_tag, rsp) }) }`."""
    h = wl.task(node)
    if handler_sleep_ms:
        h.sleep(ms=handler_sleep_ms)
    h.rpc_reply(addr, reply_code)
    s = wl.task(node, **kw)
    s.bind(addr)
    top = s.label()
    s.rpc_recv(addr, 0); s.trace(1); s.spawn(h, move_request=True); s.jmp(top)
    return s


def rpc_echo(n_clients=2, n_calls=4):
    """net/rpc.rs doc example + the shape of its users: typed `call` against a handler task, several callers in flight."""
    wl = W.WorkloadBuilder()
    ns = wl.create_node()
    asv = wl.addr(ns, 1)
    srv = _rpc_server(wl, ns, asv, 42)
    m = wl.main(); m.spawn(srv)
    cls = []
    for i in range(n_clients):
        n = wl.create_node(); a = wl.addr(n, 1)
        c = wl.task(n); c.bind(a); c.sleep(ms=10); c.set(0, n_calls)
        top = c.label()
        c.rpc_call(a, asv, 0, 5 + i); c.assert_val(42); c.djnz(0, top)
        m.spawn(c); cls.append(c)
    for c in cls:
        m.join(c)
    return wl.build()


def rpc_echo_with_data():
    """madsim/examples/rpc.rs (`Echo("hello")` -> "echo: hello") and the benches' `call_with_data` (benches/rpc.rs:28-52, net/rpc.rs:
    114-131): the request / response values and their byte payloads travel as interned (message, data) codes; the handler checks
    what it received, the caller what came back."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    hello, echoed = wl.rpc_message(("Echo", "hello")), wl.rpc_message("echo: hello")
    blob = bytes(range(256)) * 4
    put, ack = wl.rpc_message(("Put", 7), blob), wl.rpc_message("ok", blob[::-1])
    h = wl.task(ns)                                           # the per-request task: f(req, data) -> (rsp, data)
    skip = h.label() + 4
    h.jeq(put, skip); h.assert_val(hello); h.rpc_reply(asv, echoed); h.done()
    h.rpc_reply(asv, ack)
    s = wl.task(ns); s.bind(asv)
    top = s.label(); s.rpc_recv(asv, 0); s.spawn(h, move_request=True); s.jmp(top)
    c = wl.task(nc); c.bind(acl); c.sleep(ms=5)
    c.rpc_call(acl, asv, 0, hello); c.assert_val(echoed)
    c.rpc_call(acl, asv, 0, put); c.assert_val(ack)
    m = wl.main(); m.spawn(s); m.spawn(c); m.join(c)
    built = wl.build()
    assert built.rpc_messages[ack] == ("ok", blob[::-1]) and len(built.rpc_messages) == 4
    return built


def rpc_call_timeout_then_retry():
    """call_timeout (rpc.rs:96-105) elapsing while the handler is still working: the late response finds nobody holding
    its rsp_tag and stays in the caller's mailbox for good; the retry gets its own response."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    srv = _rpc_server(wl, ns, asv, 42, handler_sleep_ms=80)
    c = wl.task(nc); c.bind(acl); c.sleep(ms=10)
    c.rpc_call(acl, asv, 0, 1, timeout_ms=50); c.assert_val(A.VAL_TIMEOUT)
    c.rpc_call(acl, asv, 0, 2, timeout_ms=500); c.assert_val(42)
    c.sleep(ms=200)                                   # the first call's response has long arrived by now
    c.rpc_call(acl, asv, 0, 3); c.assert_val(42)
    m = wl.main(); m.spawn(srv); m.spawn(c); m.join(c)
    return wl.build()


def rpc_server_restart():
    """A caller keeps retrying `call_timeout` across a kill + restart of the server node (init task re-registers the handler)."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    _rpc_server(wl, ns, asv, 7, init=True, pre=True)
    c = wl.task(nc); c.bind(acl); c.sleep(ms=10); c.set(0, 12)
    top = c.label()
    c.rpc_call(acl, asv, 0, 9, timeout_ms=100)
    ok = c.label() + 3
    c.jeq(7, ok); c.flag_add(1, 1); c.jmp(ok + 1)
    assert c.label() == ok
    c.flag_add(0, 1)
    c.sleep(ms=50); c.djnz(0, top)
    c.panic_if_flag_lt(0, 4); c.panic_if_flag_lt(1, 1)    # some calls succeeded on both incarnations, some timed out
    m = wl.main(); m.spawn(c); m.sleep(ms=300); m.kill(ns); m.sleep(ms=400); m.restart(ns); m.join(c)
    return wl.build()


def rpc_hooks():
    """NetSim::hook_rpc_req / hook_rpc_rsp (net/mod.rs:240-284, consulted in NetSim::send :307-311,321-328): requests R{5}
    leaving node 1 vanish before the link test (no loss / latency draws), every response on its way to node 2 is judged
    by the hook that was installed when it was sent and dropped on arrival; replacing a hook takes effect for messages
    sent afterwards; node 3 is untouched."""
    wl = W.WorkloadBuilder()
    ns = wl.create_node(); asv = wl.addr(ns, 1)
    srv = _rpc_server(wl, ns, asv, 42)
    n1, n2, n3 = wl.create_node(), wl.create_node(), wl.create_node()
    a1, a2, a3 = wl.addr(n1, 1), wl.addr(n2, 1), wl.addr(n3, 1)
    c1 = wl.task(n1); c1.bind(a1); c1.sleep(ms=10)
    c1.rpc_call(a1, asv, 0, 5, timeout_ms=100); c1.assert_val(A.VAL_TIMEOUT)       # dropped by the request hook
    c1.rpc_call(a1, asv, 0, 6, timeout_ms=100); c1.assert_val(42)                  # another request value passes
    c2 = wl.task(n2); c2.bind(a2); c2.sleep(ms=10)
    c2.rpc_call(a2, asv, 0, 7, timeout_ms=100); c2.assert_val(A.VAL_TIMEOUT)       # the response is dropped on arrival
    c2.sleep(ms=400)                                                               # main replaces the hook meanwhile
    c2.rpc_call(a2, asv, 0, 8, timeout_ms=100); c2.assert_val(42)                  # only responses 99 are dropped now
    c3 = wl.task(n3); c3.bind(a3); c3.sleep(ms=10); c3.set(0, 3); top = c3.label()
    c3.rpc_call(a3, asv, 0, 9); c3.assert_val(42); c3.djnz(0, top)
    m = wl.main()
    m.hook_rpc_req(n1, 0, code=5); m.hook_rpc_rsp(n2)
    m.spawn(srv); m.spawn(c1); m.spawn(c2); m.spawn(c3)
    m.sleep(ms=300); m.hook_rpc_rsp(n2, code=99); m.hook_rpc_req(n1, 1)            # (a hook for another request type: R{..} of type 0 pass)
    m.join(c1); m.join(c2); m.join(c3)
    return wl.build()


ALL.update(rpc_echo_with_data=rpc_echo_with_data)
ALL.update(rpc_echo=rpc_echo, rpc_call_timeout_then_retry=rpc_call_timeout_then_retry, rpc_server_restart=rpc_server_restart,
           rpc_hooks=rpc_hooks)


def endpoint_localhost():
    """net/endpoint.rs:516-548 `localhost`, sleeps instead of the Barrier: an Endpoint bound to 127.0.0.1:1 does not
    receive what another node sends to 10.0.0.1:1 (no exact match, no 0.0.0.0:1 socket: dropped after the draws), the one
    bound to 10.0.0.1:2 receives its datagram from "10.0.0.2:1" — the sender's real IP although it bound 127.0.0.1:1 —,
    and a reply to that address finds no socket either (the sender listens on 127.0.0.1 only)."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    lo1, ip1_2, lo2 = wl.addr(n1, 1, ip="loopback"), wl.addr(n1, 2), wl.addr(n2, 1, ip="loopback")
    ip1_1 = wl.addr(n1, 1)                            # a destination only: nobody binds 10.0.0.1:1
    f1 = wl.task(n1); f1.bind(lo1); f1.bind(ip1_2)
    f1.recv_from_timeout(lo1, 1, secs=1); f1.assert_val(A.VAL_TIMEOUT)
    f1.recv_from(ip1_2, 1); f1.assert_val(1); f1.reply(ip1_2, 1, 7); f1.done()
    f2 = wl.task(n2); f2.bind(lo2); f2.sleep(ms=5)
    f2.send_to(lo2, ip1_1, 1, 1); f2.send_to(lo2, ip1_2, 1, 1)
    f2.recv_from_timeout(lo2, 1, secs=2); f2.assert_val(A.VAL_TIMEOUT); f2.done()
    m = wl.main(); m.spawn(f1); m.spawn(f2); m.join(f1); m.join(f2)
    return wl.build()


def endpoint_bind():
    """net/endpoint.rs:470-513 `bind` with named ports (ephemeral port 0 is not modelled): 0.0.0.0 and 127.0.0.1 bind on
    any node and are different keys, another node's IP is AddrNotAvailable, the node's own IP binds, binding the same
    address twice is AddrInUse — through the same table entry or through a second one naming the same address —, and a
    dropped Endpoint frees its port."""
    wl = W.WorkloadBuilder()
    n, other = wl.create_node(), wl.create_node()
    any7, lo7, foreign = wl.addr(n, 7, ip="unspecified"), wl.addr(n, 7, ip="loopback"), wl.addr(other, 9)
    ip100, ip100b = wl.addr(n, 100), wl.addr(n, 100)
    t = wl.task(n)
    t.try_bind(any7); t.assert_val(0); t.try_bind(lo7); t.assert_val(0)
    t.try_bind(foreign); t.assert_val(A.VAL_ADDR_NOT_AVAILABLE)
    t.try_bind(ip100); t.assert_val(0)
    t.try_bind(ip100); t.assert_val(A.VAL_ADDR_IN_USE); t.try_bind(ip100b); t.assert_val(A.VAL_ADDR_IN_USE)
    t.close(ip100); t.try_bind(ip100b); t.assert_val(0)          # drop and reuse port
    t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    return wl.build()


def net_wildcard_and_unbound_port():
    """network.rs:296-313: the socket lookup happens after the link test — a datagram to a port nobody listens on still
    costs the loss and latency draws and counts in msg_count —, and falls back from the exact address to 0.0.0.0:port."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, any2 = wl.addr(n1, 1), wl.addr(n2, 9, ip="unspecified")
    ip2_9, ip2_8 = wl.addr(n2, 9), wl.addr(n2, 8)              # destinations only
    r = wl.task(n2); r.bind(any2); r.recv_from(any2, 1); r.assert_val(5); r.reply(any2, 1, 6); r.done()
    s = wl.task(n1); s.bind(a1); s.sleep(ms=5)
    s.send_to(a1, ip2_8, 1, 4)                                  # unbound port: draws, msg_count, no delivery
    s.send_to(a1, ip2_9, 1, 5)                                  # no socket at 10.0.0.2:9 -> the one at 0.0.0.0:9
    s.recv_from(a1, 1); s.assert_val(6); s.done()
    m = wl.main(); m.spawn(r); m.spawn(s); m.join(r); m.join(s)
    return wl.build()


def net_ipless_node():
    """network.rs:272-290: a node created without .ip(): datagrams to 127.0.0.1 or to one of its own sockets stay on the
    node, anything else is dropped before any RNG draw ("ip not set"); other nodes cannot reach it ("destination not found")."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(ip=False), wl.create_node()
    a1, lo1, a2 = wl.addr(n1, 1), wl.addr(n1, 5, ip="loopback"), wl.addr(n2, 1)
    t1 = wl.task(n1); t1.bind(a1); t1.bind(lo1)
    t1.send_to(a1, a2, 1, 1)                                    # dropped, no draws
    t1.send_to(a1, lo1, 1, 2)                                   # loopback: delivered locally, from = 127.0.0.1:1
    t1.recv_from(lo1, 1); t1.assert_val(2)
    t1.reply(lo1, 1, 3)                                         # -> 127.0.0.1:1: no socket there (a1 is 10.0.0.1:1): dropped after the draws
    t1.recv_from_timeout(a1, 1, ms=50); t1.assert_val(A.VAL_TIMEOUT); t1.done()
    t2 = wl.task(n2); t2.bind(a2); t2.sleep(ms=5); t2.send_to(a2, a1, 1, 9)      # 10.0.0.1 is nobody's registered IP: dropped, no draws
    t2.recv_from_timeout(a2, 1, ms=50); t2.assert_val(A.VAL_TIMEOUT); t2.done()
    m = wl.main(); m.spawn(t1); m.spawn(t2); m.join(t1); m.join(t2)
    return wl.build()


def net_ipless_node_own_socket_panics():
    """network.rs:307-311: an IP-less node sending to one of its own non-loopback sockets gets as far as `.ip.unwrap()`."""
    wl = W.WorkloadBuilder()
    n1 = wl.create_node(ip=False)
    a1, b1 = wl.addr(n1, 1), wl.addr(n1, 2)
    t1 = wl.task(n1); t1.bind(a1); t1.bind(b1); t1.send_to(a1, b1, 1, 1); t1.done()
    m = wl.main(); m.spawn(t1); m.join(t1, expect_err=False)
    return wl.build()


ALL.update(endpoint_localhost=endpoint_localhost, endpoint_bind=endpoint_bind, net_wildcard_and_unbound_port=net_wildcard_and_unbound_port,
           net_ipless_node=net_ipless_node, net_ipless_node_own_socket_panics=net_ipless_node_own_socket_panics)
EXPECT_PANIC.add("net_ipless_node_own_socket_panics")


def endpoint_bind_ephemeral():
    """net/endpoint.rs:470-513 `bind`, literally (v4 cases): 0.0.0.0:0 and 127.0.0.1:0 get a non-zero port — port 1 each,
    they are different IPs —, 10.0.0.2:0 on node 1 is AddrNotAvailable, the node's own 10.0.0.1:100 binds, is dropped and
    binds again.  Then network.rs:224-236 in detail: a second 0.0.0.0:0 gets port 2, 0.0.0.0:3 named explicitly makes the
    third one skip to 4, and a dropped Endpoint's port is the next one handed out."""
    wl = W.WorkloadBuilder()
    n, other = wl.create_node(), wl.create_node()
    any_a, any_b, any_c = (wl.addr(n, 0, ip="unspecified") for _ in range(3))
    lo_a, foreign, ip100, any3 = wl.addr(n, 0, ip="loopback"), wl.addr(other, 0), wl.addr(n, 100), wl.addr(n, 3, ip="unspecified")
    t = wl.task(n)
    t.bind(any_a, port_to_val=True); t.assert_val(1)
    t.bind(lo_a, port_to_val=True); t.assert_val(1)
    t.try_bind(foreign); t.assert_val(A.VAL_ADDR_NOT_AVAILABLE)
    t.bind(ip100, port_to_val=True); t.assert_val(100); t.close(ip100); t.bind(ip100)
    t.bind(any_b, port_to_val=True); t.assert_val(2)
    t.bind(any3); t.bind(any_c, port_to_val=True); t.assert_val(4)
    t.close(any_a); t.close(any_c)
    t.bind(any_c, port_to_val=True); t.assert_val(1)            # the lowest free port again
    t.bind(any_a, port_to_val=True); t.assert_val(4)
    t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    return wl.build()


def ephemeral_clients():
    """The usual client shape: `Endpoint::bind("0.0.0.0:0")`, send to the server's address, wait for the reply the server
    sends to `from` = 10.0.0.<client>:<ephemeral port> — found through the 0.0.0.0:<port> fallback of the socket lookup
    (network.rs:304-306).  Two clients share node 2 (ports 1 and 2 in the order their binds complete, which the seed
    decides); each drops its Endpoint and binds again between requests, so ports are handed back and re-issued while
    replies to the old address are still in flight (those find nobody, or the other client's new Endpoint)."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv = wl.addr(ns, 700, ip="unspecified")
    asv_ip = wl.addr(ns, 700)                                         # what the clients dial: 10.0.0.1:700
    srv = wl.task(ns); srv.bind(asv)
    top = srv.label(); srv.recv_from(asv, 1); srv.trace_val(); srv.sleep_rand(lo_ms=0, ms=12); srv.reply(asv, 2, 0x50); srv.jmp(top)
    clients = []
    for i in range(2):
        ep = wl.addr(nc, 0, ip="unspecified")
        c = wl.task(nc); c.set(0, 4)
        top = c.label()
        c.bind(ep, port_to_val=True); c.trace_val()
        c.send_to(ep, asv_ip, 1, 0x10 + i); c.recv_from_timeout(ep, 2, ms=25); c.trace_val()
        c.close(ep); c.sleep_rand(lo_ms=0, ms=4); c.djnz(0, top); c.done()
        clients.append(c)
    m = wl.main(); m.spawn(srv)
    for c in clients:
        m.spawn(c)
    for c in clients:
        m.join(c)
    return wl.build()


def channel_wildcard_listener():
    """connect1 over general addresses (net/mod.rs:337-364): the server listens on 0.0.0.0:2379, the client binds
    0.0.0.0:0 and dials 10.0.0.1:2379.  channel(node, dst) keeps testing the link towards the DIALLED address and
    channel(dst_node, src) towards src = (client IP, client port) — both reach their sockets through the 0.0.0.0 fallback —
    so payloads flow in both directions; after the listener is dropped the client's next payload is stamped None and the
    receiver side backs off for good (the server task is gone: RESET on the client's recv)."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, dial, acl = wl.addr(ns, 2379, ip="unspecified"), wl.addr(ns, 2379), wl.addr(nc, 0, ip="unspecified")
    srv = wl.task(ns); srv.bind(asv); srv.accept1(asv); srv.set(0, 3)
    top = srv.label(); srv.chan_recv(); srv.trace_val(); srv.chan_send(0x22); srv.djnz(0, top); srv.done()
    cl = wl.task(nc); cl.bind(acl); cl.sleep(ms=10); cl.connect1(acl, dial); cl.assert_val(0); cl.set(0, 3)
    top = cl.label(); cl.chan_send(0x11); cl.chan_recv(); cl.assert_val(0x22); cl.djnz(0, top)
    cl.chan_recv(); cl.assert_val(A.VAL_RESET); cl.done()
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl)
    return wl.build()


def channel_loopback():
    """connect1 to 127.0.0.1:<port> on the client's own node: src is 127.0.0.1:<client port> (net/mod.rs:355,
    network.rs:307-311), so the server-to-client channel tests the link to a loopback address; the client Endpoint is bound
    to 0.0.0.0:0 and is found through the fallback."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    asv, dial, acl = wl.addr(n, 9, ip="unspecified"), wl.addr(n, 9, ip="loopback"), wl.addr(n, 0, ip="unspecified")
    srv = wl.task(n); srv.bind(asv); srv.accept1(asv); srv.chan_recv(); srv.assert_val(5); srv.chan_send(6); srv.sleep(ms=50); srv.done()
    cl = wl.task(n); cl.sleep(ms=5); cl.bind(acl); cl.connect1(acl, dial); cl.assert_val(0); cl.chan_send(5); cl.chan_recv(); cl.assert_val(6); cl.done()
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl); m.join(srv)
    return wl.build()


ALL.update(endpoint_bind_ephemeral=endpoint_bind_ephemeral, ephemeral_clients=ephemeral_clients,
           channel_wildcard_listener=channel_wildcard_listener, channel_loopback=channel_loopback)


def channel_self_connect_accept():
    """One task on both ends of its own connection: `ep.connect1(ep.addr)` then `ep.accept1()` (net/mod.rs:337-364,
    endpoint.rs:196-212).  The (tx, rx) the accept returns replaces the client pair the task held, so the client's Sender and
    Receiver drop at that assignment: the accepted Receiver sees the channel closed (RESET), after whatever the client end
    had sent first.  (Round-4 advisor finding: the global-state builds wrote a stale connection header back here.)"""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    a = wl.addr(n, 1)
    t = wl.task(n); t.bind(a); t.connect1(a, a); t.assert_val(0); t.accept1(a); t.chan_recv(); t.assert_val(A.VAL_RESET); t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    return wl.build()


def channel_self_connect_send_then_accept():
    """As above with a payload in flight: the client end sends 5, then the task accepts its own connection (dropping the
    client pair) and receives 5 through the accepted Receiver, then RESET; a second task dials the same Endpoint meanwhile so
    the accept queue holds two connections and the drop happens beside a live one."""
    wl = W.WorkloadBuilder()
    n, n2 = wl.create_node(), wl.create_node()
    a, b = wl.addr(n, 1), wl.addr(n2, 1)
    t = wl.task(n); t.bind(a); t.connect1(a, a); t.assert_val(0); t.chan_send(5); t.sleep(ms=20)
    t.accept1(a); t.chan_recv(); t.trace_val(); t.chan_recv(); t.trace_val()
    t.accept1(a); t.chan_recv(); t.trace_val(); t.done()
    o = wl.task(n2); o.bind(b); o.sleep(ms=5); o.connect1(b, a); o.assert_val(0); o.chan_send(9); o.sleep(ms=200); o.done()
    m = wl.main(); m.spawn(t); m.spawn(o); m.join(t); m.join(o)
    return wl.build()


ALL.update(channel_self_connect_accept=channel_self_connect_accept,
           channel_self_connect_send_then_accept=channel_self_connect_send_then_accept)


def std_system_time():
    """time/system_time.rs:122-154: `t0 = SystemTime::now(); sleep(1 s); assert!(t0.elapsed() >= 1 s); t0` — the observed
    wall-clock time depends on the seed (base time drawn around 2022), the Instant-based duration does not."""
    wl = W.WorkloadBuilder()
    m = wl.main()
    m.trace_system_time(); m.mark(); m.sleep(secs=1); m.assert_elapsed(">=", secs=1); m.trace_instant()
    return wl.build()


def getrandom_deterministic():
    """rand.rs:331-354 (issue 201): getrandom inside the simulation is served by the GlobalRng."""
    wl = W.WorkloadBuilder()
    m = wl.main()
    m.getrandom_byte(); m.trace_val(); m.random_u32(); m.trace_val()
    return wl.build()


def buggify_rates():
    """buggify.rs:36-60: of 1000 `buggify()` draws (gen_bool(0.25), config.loss_table[1]) 200..300 are true, of 1000
    `buggify_with_prob(0.1)` draws (loss_table[2]) 50..150."""
    wl = W.WorkloadBuilder()
    m = wl.main()
    for idx, flag, lo, hi in ((1, 0, 200, 300), (2, 2, 50, 150)):
        m.set(0, 1000)
        top = m.label()
        m.rand_bool(idx)
        m.jeq(0, top + 4)
        m.flag_add(flag, 1); m.jmp(top + 5)
        m.flag_add(flag + 1, 1)
        m.djnz(0, top)
        m.panic_if_flag_lt(flag, lo); m.panic_if_flag_lt(flag + 1, 1000 - hi + 1)
    return wl.build()


ALL.update(std_system_time=std_system_time, getrandom_deterministic=getrandom_deterministic, buggify_rates=buggify_rates)


def spawn_in_future_drop_by_aborting_task():
    """task/mod.rs:1184-1216: `impl Drop for A { spawn(..) }` moved into a task that is aborted before it ever runs — the task
    spawned in A::drop belongs to the same node and does run."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, spawn_on_drop=True)                      # async move { drop(a) }
    c = wl.task(n); c.flag_add(0, 1)                        # the task A::drop spawns: records that it ran (on node n)
    m = wl.main(); m.spawn(t); m.abort(t); m.join(t, expect_err=True)
    m.sleep(secs=57257); m.sleep(secs=57257); m.assert_flag(0, 1)      # 114514 s
    return wl.build()


def spawn_in_future_drop_by_killing_node():
    """task/mod.rs:1219-1253: the same guard dropped because the node was killed — spawning on the killed node succeeds, the
    new task never runs (`unreachable!()`)."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, spawn_on_drop=True)
    c = wl.task(n); c.panic(7)                              # unreachable!()
    m = wl.main(); m.spawn(t); m.kill(n); m.join(t, expect_err=True); m.sleep(secs=57257); m.sleep(secs=57257)
    return wl.build()


def spawn_in_future_drop_by_completion():
    """The guard of a body that runs to its end (`drop(a)` executed, task/mod.rs:1207): the spawned task is queued before the
    JoinHandle's awaiter is woken, so it is there when the awaiter looks."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    a = wl.addr(n, 1)
    t = wl.task(n, spawn_on_drop=True); t.bind(a); t.sleep(ms=3)
    c = wl.task(n); c.bind(a); c.flag_add(0, 1); c.sleep(ms=1); c.flag_add(0, 1)      # binds the address its parent's locals freed
    m = wl.main(); m.spawn(t); m.join(t); m.sleep(ms=10); m.assert_flag(0, 2)
    return wl.build()


def spawn_in_future_drop_across_restart():
    """A guard in an init task across a restart: the old incarnation's task is dropped under its own (killed) NodeInfo, so the
    task its guard spawns never runs; the new incarnation's init task returns, its guard spawns, and exit() kills that too."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n, init=True, spawn_on_drop=True); t.flag_add(0, 1); t.sleep(secs=2)
    c = wl.task(n); c.flag_add(1, 1)
    m = wl.main(); m.build_node(n); m.sleep(secs=1); m.restart(n); m.sleep(secs=5)
    # both incarnations started; neither guard's task ever ran: the first was spawned under the killed NodeInfo, the second just
    # before its init task's exit() killed the node (runtime/mod.rs:362-370)
    m.assert_flag(0, 2); m.assert_flag(1, 0); m.assert_exit(n, True)
    return wl.build()


def join_handle_awaits_the_task_it_named():
    """`handle.await` moves the JoinHandle into the await (task/join.rs:59-72): the awaiter gets the outcome of the task the
    handle named when the await began — here the first worker, done after 5 ms — although another task has since spawned the
    same program again (10 ms of work left) and aborted it."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    worker = wl.task(n); worker.sleep(ms=5); worker.flag_add(0, 1)
    other = wl.task(n); other.sleep(ms=1); other.spawn(worker); other.sleep(ms=1); other.abort(worker); other.sleep(ms=20)
    m = wl.main(); m.mark(); m.spawn(worker); m.spawn(other)
    m.join(worker)                                           # Ok(()): the first worker completed; the second one was cancelled
    m.assert_elapsed(">=", ms=5); m.assert_elapsed("<", ms=7); m.assert_flag(0, 1)
    m.join(worker, expect_err=True)                          # the handle variable itself now names the aborted second worker
    return wl.build()


def abort_own_handle():
    """A task that aborts its own JoinHandle (task/join.rs:158-163) keeps running until it yields — it is cancelled and woken while
    RUNNING — and is dropped when the executor pops it again: the work behind its next await never happens."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    t = wl.task(n)
    t.sleep(ms=1); t.abort(t); t.flag_add(0, 1); t.sleep(ms=5); t.flag_add(0, 10)
    m = wl.main(); m.spawn(t); m.join(t, expect_err=True); m.sleep(ms=20); m.assert_flag(0, 1)
    return wl.build()


def spawn_after_restarting_own_node():
    """task::spawn = Spawner::current() = the calling task's OWN Arc<NodeInfo> (task/mod.rs:592-599): a task that restarts its
    own node keeps running until it yields, and what it spawns meanwhile belongs to the dead incarnation — it never runs.
    The new incarnation's init task does."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    i = wl.task(n, init=True); i.flag_add(2, 1); i.sleep(secs=10)
    c = wl.task(n); c.flag_add(0, 1)
    t = wl.task(n); t.restart(n); t.spawn(c); t.flag_add(1, 1); t.sleep(ms=1); t.flag_add(1, 10)
    m = wl.main(); m.build_node(n); m.sleep(ms=5); m.spawn(t); m.sleep(secs=1)
    m.assert_flag(0, 0); m.assert_flag(1, 1); m.assert_flag(2, 2)
    return wl.build()


def spawn_after_killing_own_node():
    """The same with Handle::kill: the spawned task is created under the killed NodeInfo and dropped unpolled."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    c = wl.task(n); c.flag_add(0, 1)
    t = wl.task(n); t.kill(n); t.spawn(c); t.flag_add(1, 1); t.sleep(ms=1); t.flag_add(1, 10)
    m = wl.main(); m.spawn(t); m.sleep(secs=1); m.assert_flag(0, 0); m.assert_flag(1, 1); m.assert_exit(n, True)
    return wl.build()


ALL.update(join_handle_awaits_the_task_it_named=join_handle_awaits_the_task_it_named)
ALL.update(abort_own_handle=abort_own_handle)
ALL.update(spawn_after_restarting_own_node=spawn_after_restarting_own_node, spawn_after_killing_own_node=spawn_after_killing_own_node)
ALL.update(spawn_in_future_drop_by_aborting_task=spawn_in_future_drop_by_aborting_task,
           spawn_in_future_drop_by_killing_node=spawn_in_future_drop_by_killing_node,
           spawn_in_future_drop_by_completion=spawn_in_future_drop_by_completion,
           spawn_in_future_drop_across_restart=spawn_in_future_drop_across_restart)
CONFIGS = {"buggify_rates": dict(loss_table=(0.0, 0.25, 0.1))}


# ---- round 4: the paths the global-state builds changed (batched reads, words held in registers across a poll) ---------------

def many_endpoints_dropped_in_table_order():
    """A task that holds six Endpoints — with another task's Endpoints between them in the table, 40 table entries so the owner
    mask spans two words — completes: every Endpoint it bound is dropped (net/mod.rs:483-493 BindGuard::drop -> Network::close),
    the neighbours stay.  Afterwards the same addresses bind again, a datagram for a closed one is lost, one for a kept one arrives."""
    from madsim_amd import _abi as A
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    mine = [wl.addr(n1, 100 + 3 * i) for i in range(6)]
    theirs = [wl.addr(n1, 101 + 3 * i) for i in range(6)]
    filler = [wl.addr(n1, 200 + i) for i in range(27)]           # entries 12..38: never bound
    far = wl.addr(n1, 300)                                       # entry 39: second word of the owner mask
    a2 = wl.addr(n2, 1)
    keeper = wl.task(n1)
    for a in theirs:
        keeper.bind(a)
    keeper.bind(far)
    keeper.recv_from(theirs[2], 7); keeper.assert_val(70); keeper.recv_from(far, 8); keeper.assert_val(80); keeper.sleep(secs=5)
    holder = wl.task(n1)
    for a in mine:
        holder.bind(a)
    holder.sleep(ms=20)                                          # completes: the six go
    again = wl.task(n1)
    again.sleep(ms=200)
    for a in mine:
        again.bind(a)                                            # .unwrap(): AddrInUse would panic
    again.recv_from_timeout(mine[4], 9, ms=300); again.assert_val(A.VAL_TIMEOUT)      # the datagram sent at 100 ms was lost
    snd = wl.task(n2)
    snd.bind(a2); snd.sleep(ms=100); snd.send_to(a2, mine[4], 9, 90); snd.send_to(a2, theirs[2], 7, 70); snd.send_to(a2, far, 8, 80)
    m = wl.main()
    for t in (keeper, holder, again, snd):
        m.spawn(t)
    m.join(again); m.join(snd)
    return wl.build()


def six_connections_at_once():
    """Six clients hold a connection each at the same time (connect1's search for a free connection slot goes past the first four);
    the server's handlers answer after the last has connected; then everything is dropped and six more are made."""
    wl = W.WorkloadBuilder()
    ns = wl.create_node()
    asv = wl.addr(ns, 2379)
    handler = wl.task(ns)
    handler.chan_recv(); handler.assert_val(0x11); handler.sleep(ms=50); handler.flag_add(0, 1); handler.chan_send(0x22)
    srv = wl.task(ns)
    srv.bind(asv)
    top = srv.label()
    srv.accept1(asv); srv.spawn(handler, move_conn=True); srv.jmp(top)
    clients = []
    for i in range(6):
        nc = wl.create_node()
        acl = wl.addr(nc, 1)
        c = wl.task(nc)
        c.bind(acl); c.sleep(ms=10 + i); c.set(0, 2)
        top = c.label()
        c.connect1(acl, asv); c.assert_val(0); c.chan_send(0x11); c.chan_recv(); c.assert_val(0x22); c.chan_close(); c.sleep(ms=5); c.djnz(0, top)
        clients.append(c)
    m = wl.main()
    m.spawn(srv)
    for c in clients:
        m.spawn(c)
    for c in clients:
        m.join(c)
    m.assert_flag(0, 12)
    return wl.build()


def dead_registrations_swept_by_delivery():
    """Five timed-out receives leave five dead registrations in one mailbox (endpoint.rs:353-362: the Vec keeps them); a sixth,
    live one follows.  The next datagram sweeps the dead ones on its way (Mailbox::deliver, endpoint.rs:331-351) and is taken by
    the live receive; a second datagram with another tag finds its own registration behind them."""
    from madsim_amd import _abi as A
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    r = wl.task(n1)
    r.bind(a1)
    for _ in range(5):
        r.recv_from_timeout(a1, 1, ms=10); r.assert_val(A.VAL_TIMEOUT)
    r.recv_from_timeout(a1, 1, secs=2); r.assert_val(41)
    r.recv_from_timeout(a1, 1, ms=10); r.assert_val(A.VAL_TIMEOUT); r.recv_from_timeout(a1, 1, ms=10); r.assert_val(A.VAL_TIMEOUT)
    r.recv_from(a1, 1); r.assert_val(43)
    other = wl.task(n1)
    other.sleep(ms=5); other.recv_from(a1, 2); other.assert_val(42)
    s = wl.task(n2)
    s.bind(a2); s.sleep(ms=200); s.send_to(a2, a1, 1, 41); s.sleep(ms=100); s.send_to(a2, a1, 2, 42); s.sleep(ms=100); s.send_to(a2, a1, 1, 43)
    m = wl.main()
    m.spawn(r); m.spawn(other); m.spawn(s); m.join(r); m.join(other)
    return wl.build()


ALL.update(many_endpoints_dropped_in_table_order=many_endpoints_dropped_in_table_order, six_connections_at_once=six_connections_at_once,
           dead_registrations_swept_by_delivery=dead_registrations_swept_by_delivery)


# ---- round 6: NetSim::update_config of send_latency (net/mod.rs:138-141, network.rs:129, consumed at :267) ----

def update_config_latency():
    """`NetSim::current().update_config(|c| c.send_latency = lo..hi)` between datagrams: every link test samples the range in force at
    that moment (network.rs:267), a message in flight keeps the latency it drew.  The table: 100..101 ms (UniformDuration's Small path,
    A.3), 1.5..3.5 s (Medium path: the range crosses a seconds boundary), 1..2 ns (exactly 1 ns).  The receiver times every arrival
    from its own t0; with the default 1..10 ms in force throughout every assert but the first would fail."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a_tx, a_rx = wl.addr(n1, 1), wl.addr(n2, 1)
    rx = wl.task(n2); rx.mark(); rx.bind(a_rx)
    rx.recv_from(a_rx, 1); rx.assert_val(0xA); rx.assert_elapsed(">=", ms=12); rx.assert_elapsed("<", ms=25); rx.trace_instant()
    rx.recv_from(a_rx, 2); rx.assert_val(0xB); rx.assert_elapsed(">=", ms=110); rx.assert_elapsed("<", ms=120); rx.trace_instant()
    rx.recv_from(a_rx, 3); rx.assert_val(0xC); rx.assert_elapsed(">=", secs=1, ms=500); rx.assert_elapsed("<", secs=3, ms=600); rx.trace_instant()
    rx.recv_from(a_rx, 4); rx.assert_val(0xD); rx.assert_elapsed(">=", secs=5, ms=2); rx.assert_elapsed("<", secs=5, ms=4); rx.trace_instant()
    tx = wl.task(n1); tx.mark(); tx.bind(a_tx); tx.sleep(ms=10)
    tx.send_to(a_tx, a_rx, 1, 0xA)
    tx.set_latency(0); tx.send_to(a_tx, a_rx, 2, 0xB)
    tx.set_latency(1); tx.send_to(a_tx, a_rx, 3, 0xC)
    tx.sleep_until(secs=5); tx.set_latency(2); tx.send_to(a_tx, a_rx, 4, 0xD)
    m = wl.main(); m.spawn(rx); m.spawn(tx); m.join(rx); m.join(tx)
    return wl.build()


def update_config_latency_channel():
    """The same switch seen by the reliable channel: connect1's link test draws (and discards) a latency, `tx.send` stamps the payload
    with now + latency (net/mod.rs:417-421, 375-380) and the receiver sleeps until then (:399).  The supervisor — a different task on a
    different node: the config is the simulator's, not the caller's — sets 200..201 ms before the second payload and the launch's own
    default never comes back."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    srv = wl.task(ns); srv.bind(asv); srv.accept1(asv)
    srv.chan_recv(); srv.assert_val(1); srv.mark()
    srv.chan_recv(); srv.assert_val(2); srv.assert_elapsed(">=", ms=280); srv.assert_elapsed("<", ms=305); srv.trace_instant()
    srv.chan_recv(); srv.assert_val(3); srv.assert_elapsed("<", ms=310); srv.trace_instant()       # sent right behind it, same range: ordered delivery
    cl = wl.task(nc); cl.bind(acl); cl.sleep(ms=10); cl.connect1(acl, asv); cl.assert_val(0); cl.chan_send(1)
    cl.sleep(ms=100); cl.chan_send(2); cl.chan_send(3); cl.sleep(secs=1)
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.sleep(ms=60); m.set_latency(0); m.join(srv)
    return wl.build()


ALL.update(update_config_latency=update_config_latency, update_config_latency_channel=update_config_latency_channel)
CONFIGS["update_config_latency"] = dict(lat_table=((100_000_000, 101_000_000), (1_500_000_000, 3_500_000_000), (1, 2)))
CONFIGS["update_config_latency_channel"] = dict(lat_table=((200_000_000, 201_000_000),))


# ---- MADSIM_STATE_NARROW_HEAP: the two run-time conditions that send a seed back to the 16-byte entries (k_timer.h timer_add) -----------

def narrow_heap_backoff_past_the_horizon():
    """-> (workload, limits).  A channel receiver whose link stays clogged for 5.7 s: its back-off doubles to 4 096 ms (net/mod.rs:388-398), a
    deadline the 8-byte entries' 2^31 ns horizon cannot hold — nothing in the table says so (every sleep is below 2 s).  Every op class is
    present (a timeout, a typed call, a kill) so that the build is the one that carries the narrow-heap variant."""
    wl = W.WorkloadBuilder()
    ns, nc, nx = wl.create_node(), wl.create_node(), wl.create_node()
    asv, acl, ax = wl.addr(ns, 1), wl.addr(nc, 1), wl.addr(nx, 1)
    srv = wl.task(ns); srv.bind(asv); srv.accept1(asv); srv.mark(); srv.chan_recv(); srv.assert_val(7); srv.assert_elapsed(">=", secs=5); srv.trace_instant()
    cl = wl.task(nc); cl.bind(acl); cl.sleep(ms=10); cl.connect1(acl, asv); cl.assert_val(0); cl.sleep(ms=100); cl.chan_send(7); cl.sleep(secs=1)
    idle = wl.task(nx); idle.bind(ax); idle.recv_from_timeout(ax, 1, ms=20); idle.rpc_call(ax, asv, 0, 1, timeout_ms=5)
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.spawn(idle); m.sleep(ms=50); m.clog_link(nc, ns); m.sleep(ms=1900); m.sleep(ms=1900); m.sleep(ms=1900)
    m.kill(nx); m.unclog_link(nc, ns); m.join(srv)
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 30; lim.mbox_regs, lim.mbox_msgs = 4, 4
    return wl.build(), lim


def narrow_heap_forty_datagrams_in_flight():
    """-> (workload, config, limits).  Ten senders x four datagrams under a 400-800 ms latency: forty deliveries in flight at once, more than the
    32 records a small heap's pool holds."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    rx = wl.task(n2); rx.bind(a2); rx.set(0, 40); top = rx.label(); rx.recv_from_timeout(a2, 1, ms=900); rx.trace_val(); rx.djnz(0, top)
    senders = []
    for k in range(10):
        t = wl.task(n1); t.sleep(ms=5)
        for _ in range(4):
            t.send_to(a1, a2, 1, 0x100 + k)
        senders.append(t)
    b = wl.task(n1); b.bind(a1); b.sleep(secs=1)
    m = wl.main(); m.spawn(b); m.spawn(rx); m.sleep(ms=3)
    for t in senders:
        m.spawn(t)
    m.join(rx)
    cfg = A.Config.default(lat_lo_ns=400_000_000, lat_hi_ns=800_000_000)
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 4, 60; lim.mbox_regs, lim.mbox_msgs = 4, 48; lim.max_tasks = 16
    return wl.build(), cfg, lim


def model_ceiling_workloads():
    """One workload per ceiling of the device runner's workload MODEL (include/madsim_hip.h MADSIM_UNSUPPORTED): (name, workload, limits,
    the oracle's event bit — oracle/madsim_oracle.h MADSIM_ORACLE_ME_*).  The reference itself has none of these ceilings."""
    from madsim_amd import _abi as A
    ws = []
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    v = wl.virtual_addr(1, 80); a = wl.addr(n, 1)
    svc = wl.ipvs_service(v, [a] * 6)
    t = wl.task(n); t.sleep(ms=1); t.ipvs_add_server(svc, a)
    m = wl.main(); m.spawn(t); m.join(t)
    ws.append(("ipvs_seventh_server", wl.build(), None, 32))
    wl = W.WorkloadBuilder()                               # nine clients dial one listener that never accepts
    ns, nc = wl.create_node(), wl.create_node()
    asv = wl.addr(ns, 1)
    srv = wl.task(ns); srv.bind(asv); srv.sleep(secs=5); srv.done()
    cls = []
    for i in range(9):
        ac = wl.addr(nc, 2 + i)
        c = wl.task(nc); c.bind(ac); c.sleep(ms=10); c.connect1(ac, asv); c.sleep(secs=1); c.done(); cls.append(c)
    m = wl.main(); m.spawn(srv)
    for c in cls: m.spawn(c)
    for c in cls: m.join(c)
    lim = A.Limits(); lim.max_conns, lim.max_tasks = 16, 16
    ws.append(("ninth_queued_connection", wl.build(), lim, 16))
    wl = W.WorkloadBuilder()                               # panic!("{}", flag + 7) with flag = 300 > panic_dyn_max
    n = wl.create_node(restart_on_panic_matching=("1",))
    t = wl.task(n); t.flag_add(0, 300); t.panic_with_flag(0, 7)
    m = wl.main(); m.spawn(t); m.join(t, expect_err=True)
    ws.append(("panic_dyn_beyond_max", wl.build(), None, 64))
    wl = W.WorkloadBuilder()                               # a spawn storm: 300 sleepers alive at once (the layout's 8-bit task slot holds 254)
    n = wl.create_node()
    sl = wl.task(n); sl.sleep(secs=1); sl.done()
    m = wl.main(); m.set(0, 300); top = m.label(); m.spawn(sl); m.djnz(0, top); m.sleep(secs=2); m.done()
    lim = A.Limits(); lim.max_tasks = 254
    ws.append(("spawn_storm_255_tasks", wl.build(), lim, 1))
    wl = W.WorkloadBuilder()                               # 3 x 86 timed-out receives on one socket: 255 dead registrations stay, the 256th has no room
    n = wl.create_node()
    a = wl.addr(n, 1)
    ts = []
    for i in range(3):
        t = wl.task(n)
        if i == 0: t.bind(a)
        else: t.sleep(us=300 * i)
        t.set(0, 86); top = t.label(); t.recv_from_timeout(a, 1, ms=1); t.djnz(0, top); t.sleep(secs=1); t.done(); ts.append(t)
    m = wl.main()
    for t in ts: m.spawn(t)
    for t in ts: m.join(t)
    lim = A.Limits(); lim.mbox_regs = 255
    ws.append(("registrations_256", wl.build(), lim, 2))
    wl = W.WorkloadBuilder()                               # ONE task, 129 timed-out receives: its 8-bit receive sequence number wraps onto a dead registration
    n = wl.create_node()
    a = wl.addr(n, 1)
    t = wl.task(n); t.bind(a); t.set(0, 129); top = t.label(); t.recv_from_timeout(a, 1, ms=1); t.djnz(0, top); t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    lim = A.Limits(); lim.mbox_regs = 255
    ws.append(("rxseq_wraps_onto_dead_registration", wl.build(), lim, 4))
    wl = W.WorkloadBuilder()                               # 16 payloads queued in one channel direction (the receiver never receives)
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    srv = wl.task(ns); srv.bind(asv); srv.accept1(asv); srv.sleep(secs=30); srv.done()
    cl = wl.task(nc); cl.bind(acl); cl.sleep(ms=10); cl.connect1(acl, asv); cl.set(0, 16); top = cl.label(); cl.chan_send(7); cl.djnz(0, top); cl.done()
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl)
    lim = A.Limits(); lim.chan_queue = 15
    ws.append(("sixteenth_queued_payload", wl.build(), lim, 8))

    wl = W.WorkloadBuilder()                               # 256 datagrams for a bound socket nobody receives from: the 256th has no room in the model's mailbox
    n1, n2 = wl.create_node(), wl.create_node()
    a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    rx = wl.task(n2); rx.bind(a2); rx.recv_from_timeout(a2, 9, ms=1); rx.sleep(secs=30); rx.done()      # (an extended op: mailboxes of 255)
    tx = wl.task(n1); tx.bind(a1); tx.sleep(ms=5); tx.set(0, 256); top = tx.label(); tx.send_to(a1, a2, 1, 9); tx.djnz(0, top); tx.sleep(ms=50); tx.done()
    m = wl.main(); m.spawn(rx); m.spawn(tx); m.join(tx)
    lim = A.Limits(); lim.mbox_msgs = 255
    ws.append(("queued_messages_256", wl.build(), lim, 4096))
    return ws


def config(name):
    """Non-default Config a workload is meant to run under (None = Config::default())."""
    return A.Config.default(**CONFIGS[name]) if name in CONFIGS else None


def limits(name):
    """Device capacities a workload needs beyond the defaults (None = defaults)."""
    if name in ("rpc_server_restart", "rpc_hooks", "ipvs_service_lifecycle"):   # timed-out calls / receives leave dead registrations behind (rpc.rs:125)
        lim = A.Limits(); lim.mbox_regs, lim.mbox_msgs = 8, 4
        return lim
    if name == "ipvs_round_robin_datagrams":             # four replies may queue up before the sender starts receiving
        lim = A.Limits(); lim.mbox_msgs = 6
        return lim
    if name == "six_connections_at_once":
        lim = A.Limits(); lim.max_conns, lim.max_tasks = 8, 16
        return lim
    if name == "dead_registrations_swept_by_delivery":
        lim = A.Limits(); lim.mbox_regs, lim.mbox_msgs = 12, 4
        return lim
    if name == "many_endpoints_dropped_in_table_order":
        lim = A.Limits(); lim.max_tasks = 8
        return lim
    if name == "kill_many_tasks":                        # 13 sleepers + the init task, twice over while a restart's kill is pending
        lim = A.Limits(); lim.max_tasks = 40
        return lim
    if name == "join_handle_awaits_the_task_it_named":   # two instances of one program alive at once
        lim = A.Limits(); lim.max_tasks = 6
        return lim
    return None


# ---- IP Virtual Server (net/ipvs.rs; consulted by NetSim::send and connect1, net/mod.rs:312-317,345-350) --------------------

def ipvs_load_balance():
    """net/tcp/mod.rs:254-315 `ipvs_load_balance` over the Endpoint layer the TCP sim rides on (connect1 / accept1): service
    1.1.1.1:80 -> [10.0.0.1:1, 10.0.0.2:1], listeners on 0.0.0.0:1 of nodes 1 and 2, node 3 connects to 1.1.1.1:80 twice —
    the first connection reaches node 1, the second node 2 — and writes "1" / "2"; each listener asserts what it reads.
    (sleeps order the set-up where the reference uses a tokio Barrier.)"""
    wl = W.WorkloadBuilder()
    n1, n2, n3 = wl.create_node(), wl.create_node(), wl.create_node()
    l1, l2 = wl.addr(n1, 1, ip="unspecified"), wl.addr(n2, 1, ip="unspecified")
    s1, s2 = wl.addr(n1, 1), wl.addr(n2, 1)                       # the real servers' addresses, as add_server names them
    vip = wl.virtual_addr(1, 80)
    wl.ipvs_service(vip, [s1, s2])
    c = wl.addr(n3, 0, ip="unspecified")                          # TcpStream::connect binds an ephemeral Endpoint
    f1 = wl.task(n1); f1.bind(l1); f1.accept1(l1); f1.chan_recv(); f1.assert_val(wl.payload(b"1"))
    f2 = wl.task(n2); f2.bind(l2); f2.accept1(l2); f2.chan_recv(); f2.assert_val(wl.payload(b"2"))
    hold = wl.task(n3); hold.chan_send(wl.payload(b"1")); hold.sleep(ms=200)       # stream1 lives on in its own task
    f3 = wl.task(n3); f3.sleep(ms=50); f3.bind(c)
    f3.connect1(c, vip); f3.assert_val(0); f3.spawn(hold, move_conn=True)          # go to node 1
    f3.connect1(c, vip); f3.assert_val(0); f3.chan_send(wl.payload(b"2")); f3.sleep(ms=200)   # go to node 2
    m = wl.main(); m.spawn(f1); m.spawn(f2); m.spawn(f3); m.join(f1); m.join(f2); m.join(f3)
    return wl.build()


def ipvs_round_robin_datagrams():
    """ipvs.rs:88-105 get_server over datagrams: five send_to(1.1.1.1:80) go to the three servers in turn and wrap
    (a, b, c, a, b); rr_index advances also when the chosen server has no socket (server c: nobody listens — the datagram is
    lost after its draws); a datagram to a virtual address WITHOUT a service (1.1.1.2:80) is dropped before any draw; one to
    a service with no servers likewise; replies go back to the sender's real address."""
    wl = W.WorkloadBuilder()
    n1, n2, n3, n4 = wl.create_node(), wl.create_node(), wl.create_node(), wl.create_node()
    a, b, cdead = wl.addr(n1, 1), wl.addr(n2, 1), wl.addr(n3, 1)
    vip, lonely, empty = wl.virtual_addr(1, 80), wl.virtual_addr(2, 80), wl.virtual_addr(3, 80)
    wl.ipvs_service(vip, [a, b, cdead]); wl.ipvs_service(empty, [])
    me = wl.addr(n4, 1)
    ra = wl.task(n1); ra.bind(a); ra.set(0, 2); top = ra.label(); ra.recv_from(a, 1); ra.trace_val(); ra.reply(a, 2, 100); ra.djnz(0, top)
    rb = wl.task(n2); rb.bind(b); rb.set(0, 2); top = rb.label(); rb.recv_from(b, 1); rb.trace_val(); rb.reply(b, 2, 200); rb.djnz(0, top)
    tx = wl.task(n4); tx.bind(me); tx.sleep(ms=20)
    for k in range(5):
        tx.send_to(me, vip, 1, 10 + k); tx.sleep(ms=15)
    tx.send_to(me, lonely, 1, 77); tx.send_to(me, empty, 1, 78)
    tx.set(0, 4); top = tx.label(); tx.recv_from(me, 2); tx.trace_val(); tx.djnz(0, top)
    tx.recv_from_timeout(me, 2, ms=100); tx.assert_val(A.VAL_TIMEOUT)
    m = wl.main(); m.spawn(ra); m.spawn(rb); m.spawn(tx); m.join(ra); m.join(rb); m.join(tx)
    return wl.build()


def ipvs_rpc_call_asserts_given_address():
    """rpc.rs:126 `assert_eq!(from, dst)`: a typed call through a virtual address reaches the real server, whose response
    comes from ITS address, not from the one the caller was given — the caller panics, as in the reference."""
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, vip, me = wl.addr(ns, 1), wl.virtual_addr(1, 80), wl.addr(nc, 1)
    wl.ipvs_service(vip, [asv])
    srv = _rpc_server(wl, ns, asv, 42)
    cl = wl.task(nc); cl.bind(me); cl.sleep(ms=10); cl.rpc_call(me, vip, 0, 5); cl.assert_val(42)
    m = wl.main(); m.spawn(srv); m.spawn(cl); m.join(cl)
    return wl.build()


def _ipvs_receivers(wl, nodes, counts, reply=False):
    """One listener per node on 10.0.0.<node>:1 that receives `counts[i]` datagrams of tag 1 and folds each payload into the
    observation hash (so WHO received WHAT is part of the result)."""
    addrs, tasks = [], []
    for n, cnt in zip(nodes, counts):
        a = wl.addr(n, 1)
        t = wl.task(n); t.bind(a)
        if cnt:
            t.set(0, cnt); top = t.label(); t.recv_from(a, 1); t.trace_val()
            if reply:
                t.reply(a, 2, 0x500 + n)
            t.djnz(0, top)
        addrs.append(a); tasks.append(t)
    return addrs, tasks


def ipvs_round_robin_unit_test():
    """net/ipvs.rs:113-131 `round_robin`, call for call, with get_server observed through NetSim::send (net/mod.rs:312-317):
    add_service; get_server -> None (the datagram keeps its virtual destination and is dropped before any draw); three
    add_server; three get_server -> servers 1, 2, 3; del_server(server 1); get_server -> server 2 (rr_index 3 >= len 2 wraps to
    0, ipvs.rs:96-98).  Receiver 1 asserts it got datagram 1 only, receiver 2 datagrams 2 and 4, receiver 3 datagram 3."""
    wl = W.WorkloadBuilder()
    n1, n2, n3, n4 = (wl.create_node() for _ in range(4))
    vip = wl.virtual_addr(1, 80)
    svc = wl.ipvs_service(vip, absent=True)
    s1, s2, s3 = wl.addr(n1, 1), wl.addr(n2, 1), wl.addr(n3, 1)
    r1 = wl.task(n1); r1.bind(s1); r1.recv_from(s1, 1); r1.assert_val(1); r1.recv_from_timeout(s1, 1, ms=200); r1.assert_val(A.VAL_TIMEOUT)
    r2 = wl.task(n2); r2.bind(s2); r2.recv_from(s2, 1); r2.assert_val(2); r2.recv_from(s2, 1); r2.assert_val(4)
    r3 = wl.task(n3); r3.bind(s3); r3.recv_from(s3, 1); r3.assert_val(3); r3.recv_from_timeout(s3, 1, ms=200); r3.assert_val(A.VAL_TIMEOUT)
    me = wl.addr(n4, 1)
    op = wl.task(n4); op.bind(me); op.sleep(ms=10)
    op.ipvs_add_service(svc)
    op.send_to(me, vip, 1, 99); op.sleep(ms=20)                   # get_server -> None
    op.ipvs_add_server(svc, s1); op.ipvs_add_server(svc, s2); op.ipvs_add_server(svc, s3)
    for k in (1, 2, 3):
        op.send_to(me, vip, 1, k); op.sleep(ms=20)
    op.ipvs_del_server(svc, s1)
    op.send_to(me, vip, 1, 4); op.sleep(ms=20)
    m = wl.main()
    for t in (r1, r2, r3, op):
        m.spawn(t)
    for t in (r1, r2, r3, op):
        m.join(t)
    return wl.build()


def ipvs_add_server_without_service():
    """ipvs.rs:67-75 `.expect("service not found")`: add_server on a service nobody added panics the calling task."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    vip = wl.virtual_addr(1, 80)
    svc = wl.ipvs_service(vip, absent=True)
    s1 = wl.addr(n1, 1)
    op = wl.task(n2); op.sleep(ms=1); op.ipvs_add_server(svc, s1)
    m = wl.main(); m.spawn(op); m.join(op)
    return wl.build()


def ipvs_del_server_without_service():
    """ipvs.rs:78-85: del_server after del_service panics too (the service left the HashMap with its servers)."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    vip = wl.virtual_addr(1, 80)
    s1 = wl.addr(n1, 1)
    svc = wl.ipvs_service(vip, [s1])
    op = wl.task(n2); op.sleep(ms=1); op.ipvs_del_service(svc); op.ipvs_del_server(svc, s1)
    m = wl.main(); m.spawn(op); m.join(op)
    return wl.build()


def ipvs_service_lifecycle():
    """del_service stops the rewrite (the datagram is dropped: nobody owns the virtual address); add_service on an EXISTING
    service is HashMap::insert of a fresh one — servers gone, rr_index 0; del_server removes every equal address (retain);
    an rr_index left beyond the shorter list wraps to its head.  Three receivers reply with their own code, the operator
    traces each reply (or the timeout)."""
    wl = W.WorkloadBuilder()
    n1, n2, n3, n4 = (wl.create_node() for _ in range(4))
    vip = wl.virtual_addr(1, 80)
    (a, b, c), rx = _ipvs_receivers(wl, (n1, n2, n3), (0, 0, 0))
    for t, addr, code in zip(rx, (a, b, c), (0xA, 0xB, 0xC)):      # serve until the operator is done
        top = t.label(); t.recv_from_timeout(addr, 1, ms=400)
        done = t.label() + 3
        t.jeq(A.VAL_TIMEOUT, done); t.reply(addr, 2, code); t.jmp(top)
        assert t.label() == done
    svc = wl.ipvs_service(vip, [a, b, c])
    me = wl.addr(n4, 1)
    op = wl.task(n4); op.bind(me); op.sleep(ms=5)

    def probe():
        op.send_to(me, vip, 1, 7); op.recv_from_timeout(me, 2, ms=30); op.trace_val()
    probe(); probe(); probe()                                     # a b c: rr_index = 3
    op.ipvs_del_server(svc, c); probe()                            # [a, b], 3 >= 2 -> a
    op.ipvs_add_server(svc, a); op.ipvs_add_server(svc, c)         # [a, b, a, c], rr_index 1
    probe(); probe(); probe()                                     # b a c
    op.ipvs_del_server(svc, a); probe()                            # retain: [b, c]; rr_index 4 -> b
    op.ipvs_del_service(svc); probe()                              # no service: timeout
    op.ipvs_add_service(svc); probe()                              # a fresh service without servers: timeout
    op.ipvs_add_server(svc, c); probe(); probe()                   # c c
    op.ipvs_add_service(svc); probe()                              # insert again: servers gone
    m = wl.main()
    for t in rx + [op]:
        m.spawn(t)
    m.join(op)
    return wl.build()


ALL.update(ipvs_load_balance=ipvs_load_balance, ipvs_round_robin_datagrams=ipvs_round_robin_datagrams,
           ipvs_rpc_call_asserts_given_address=ipvs_rpc_call_asserts_given_address,
           ipvs_round_robin_unit_test=ipvs_round_robin_unit_test, ipvs_add_server_without_service=ipvs_add_server_without_service,
           ipvs_del_server_without_service=ipvs_del_server_without_service, ipvs_service_lifecycle=ipvs_service_lifecycle)
EXPECT_PANIC.update(("ipvs_rpc_call_asserts_given_address", "ipvs_add_server_without_service", "ipvs_del_server_without_service"))


# ---- MADSIM_STATE_DEDUP_TIMERS (k_timer.h dedup_note): not in ALL — these two exist to exercise that switch ------------------

def timeout_repeats_and_ties(pairs=5, rounds=40, chatter=30):
    """A timeout-only workload (the build the switch applies to) with both things the de-duplicated timer heap must get right:
    * repeats — `timeout(20 ms, ep.recv_from(tag))` completing through a message: the poll that takes the message registers the
      timeout's Sleep AGAIN (time/sleep.rs:51-53), both copies fire 20 ms later as wake-ups of a task that has moved on, whose
      spurious poll re-registers whatever Sleep it is in by then;
    * ties between DIFFERENT events — pairs of tasks looping `sleep(1 ms + 75 ns)` / `sleep(1 ms)`: polled back to back the two
      deadlines coincide whenever the 50..100 ns poll cost in between (task/mod.rs:319-321) comes out at 75, and which of them
      fires first is decided by the BinaryHeap's array order; each traces its id after waking, so the order is observable."""
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    a_rx, a_tx = wl.addr(n1, 1), wl.addr(n2, 1)
    rx = wl.task(n1); rx.bind(a_rx); rx.set(0, chatter)
    top = rx.label(); rx.recv_from_timeout(a_rx, 7, ms=20); rx.trace_val(); rx.djnz(0, top); rx.done()
    tx = wl.task(n2); tx.bind(a_tx); tx.set(0, chatter - 2)
    top = tx.label(); tx.sleep(ms=3); tx.send_to(a_tx, a_rx, 7, 0x55); tx.djnz(0, top); tx.done()
    tasks = [rx, tx]
    for p in range(pairs):
        a = wl.task(n1); a.set(0, rounds); top = a.label(); a.sleep(ms=1, ns=75); a.trace(0x100 + 2 * p); a.djnz(0, top); a.done()
        b = wl.task(n2); b.set(0, rounds); top = b.label(); b.sleep(ms=1); b.trace(0x101 + 2 * p); b.djnz(0, top); b.done()
        tasks += [a, b]
    m = wl.main()
    for t in tasks:
        m.spawn(t)
    for t in tasks:
        m.join(t)
    return wl.build()


def dedup_limits(base=None):
    """Global-state layout with the re-registered Sleep timers kept as counts (MADSIM_STATE_DEDUP_TIMERS)."""
    import copy
    lim = copy.copy(base) if base is not None else A.Limits()
    if lim.lanes_per_wave != 32:            # (the election loop's limits ask for 32 seed lanes per wave on the global-state build: kept)
        lim.lanes_per_wave = 0
    lim.state_mem = A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS
    return lim


def kill_many_tasks(n_sleepers=13):
    """NodeInfo::kill marks and wakes every task of the node IN SPAWN ORDER (task/mod.rs:133-140): a node with more tasks than the global-state
    kernel's kill keeps per scan (four: k_lifecycle.h info_kill), killed, restarted twice (the second restart kills the first's tasks) and
    killed again, with the tasks' timers still in the heap."""
    wl = W.WorkloadBuilder()
    n = wl.create_node()
    ts = []
    for i in range(n_sleepers):
        t = wl.task(n); top = t.label(); t.sleep(ms=3 + i); t.flag_add(0, 1); t.jmp(top); ts.append(t)
    ini = wl.task(n, init=True)
    for t in ts:
        ini.spawn(t)
    top = ini.label(); ini.sleep(ms=50); ini.jmp(top)
    m = wl.main()
    m.build_node(n)
    m.sleep(ms=20); m.kill(n); m.sleep(ms=5); m.restart(n); m.sleep(ms=17); m.restart(n); m.sleep(ms=9); m.kill(n); m.sleep(ms=3)
    return wl.build()


ALL.update(kill_many_tasks=kill_many_tasks)
