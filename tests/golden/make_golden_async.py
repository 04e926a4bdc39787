#!/usr/bin/env python3
"""Generates tests/golden/executor_kat_async.json — known answers for the oracle's await-heavy ops.

A THIRD formulation of the executor, independent of both oracle/madsim_oracle.c (sub-state machine in C) and
make_golden.py (sub-state machine in Python): here every task body is a Python *generator* — `yield` is
`Poll::Pending`, `generator.close()` is dropping the future (its `finally:` blocks are the `Drop` impls), a
`timeout()` is literally `select_biased!` over an inner generator and a Sleep.  Written from the reference
sources: time/mod.rs:103-140 (sleep, timeout), time/sleep.rs:47-54 (Sleep::poll registers a timer on every
not-elapsed poll), net/endpoint.rs:120-149,331-362 (send_to_raw / recv_from_raw / Mailbox), net/mod.rs:287-333
(rand_delay, send), net/network.rs:162-203,261-313 (clog sets, try_send), net/rpc.rs:96-180 (call, call_timeout,
add_rpc_handler), task/mod.rs:220-323 (executor loop), rand.rs:64-88,142-158 (log, RngCore).  It shares only the
generator / gen_range / UniformDuration arithmetic with make_golden.py.  Round 2 added the network's address handling —
sockets live in a literal per-node `HashMap<(ip, port), socket>`, `Network::bind` / `resolve_dest_node` / `try_send`
(net/network.rs:206-313) are restated over it with addresses as (ip string, port) tuples, the receiver's `from` is the
tuple the reference builds (:307-311) — the NetSim request / response hooks (net/mod.rs:240-284,307-328), ephemeral ports
(network.rs:224-236) and the reliable channel (net/mod.rs:337-430, endpoint.rs:178-262): connect1 / accept1 / Sender / Receiver
with explicit Arc<BindGuard> counts, the async_channel accept queue and the receiver's backoff stream.

Like make_golden.py it cannot be pinned to the Rust reference in this image; what the fixture gives is agreement of
independently written restatements on timeouts' duplicate timers, dropped receivers, orphaned RPC responses.

Run:  python tests/golden/make_golden_async.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from madsim_amd import _abi as A  # noqa: E402
from madsim_amd import workload as W  # noqa: E402
from make_golden import FNV_OFFSET, FNV_PRIME, M64, OPN, Xoshiro, duration_params, gen_range_attempts  # noqa: E402

MS = 1_000_000


class Panic(Exception):
    """panic!(..): `code` is what NodeBuilder::restart_on_panic_matching patterns are compared with (255 = any other message)."""

    def __init__(self, code=255):
        super().__init__(code)
        self.code = code


class NodeInfo:
    """task/mod.rs:86-110: one per node incarnation; Handle::restart replaces it."""

    def __init__(self, node):
        self.node, self.killed, self.paused, self.tasks = node, False, False, []

    def kill(self, sim):                            # NodeInfo::kill (:133-140)
        self.killed = True
        tasks, self.tasks = self.tasks, []          # drain(..)
        for t in tasks:
            if t.alive:                             # Weak<TaskInfo>::upgrade: the TaskInfo lives as long as the future
                sim.wake(t)


class Oneshot:
    """tokio::sync::oneshot as the mailbox uses it: one value, a receiver that may be dropped."""

    def __init__(self):
        self.val, self.rx_alive, self.rx_task = None, True, None


class Arc:
    """Arc<T> with an explicit strong count; `on_zero` is T's Drop impl."""

    def __init__(self, on_zero):
        self.n, self.on_zero = 1, on_zero

    def clone(self):
        self.n += 1
        return self

    def drop(self):
        self.n -= 1
        if self.n == 0:
            self.on_zero()


class Mpsc:
    """tokio::sync::mpsc::unbounded_channel as channel() uses it: one sender, one receiver."""

    def __init__(self):
        self.q, self.tx_alive, self.rx_alive, self.rx_task = [], True, True, None


class Task:
    def __init__(self, sim, prog, info):
        self.prog, self.node, self.info = prog, sim.progs[prog][0], info
        self.cancelled, self.outcome = False, None  # TaskInfo.cancelled; "completed" / "cancelled" once the future is gone
        self.alive, self.sched, self.running, self.joiner = True, True, False, None
        self.cnt, self.val, self.frm, self.aux, self.t0 = [0, 0], 0, 0, 0, 0
        self.owned = []
        self.guard_gone = False
        self.conn = None                            # the (Sender, Receiver) pair this task holds (one at a time: the VM's rule)
        self.gen = sim.body(self, sim.progs[prog][2])


class Sim:
    def __init__(self, w, cfg, seed):
        self.insns = [(w.insns[i].op, w.insns[i].a, w.insns[i].b, w.insns[i].imm) for i in range(w.struct.n_insns)]
        self.progs = [(w.progs[i].node, w.progs[i].flags, w.progs[i].entry) for i in range(w.struct.n_progs)]
        self.socks = [(w.socks[i].node, w.socks[i].port) for i in range(w.struct.n_socks)]
        # SocketAddr of table entry i, and the network's view of the nodes (network.rs:19-37)
        self.addr = [({0: "10.0.0.%d" % w.socks[i].node, 1: "0.0.0.0", 2: "127.0.0.1", 3: "1.1.1.%d" % w.socks[i].node}[w.socks[i].kind],
                      w.socks[i].port) for i in range(w.struct.n_socks)]
        # IpVirtualServer (net/ipvs.rs): HashMap<ServiceAddr, Service { servers: Vec<String>, rr_index }>, filled before block_on
        self.ipvs = {}
        for k in range(w.struct.n_services):
            sv = w.services[k]
            if sv.n_servers & 0x80:                 # declared only: a task adds the service (MS_OP_IPVS)
                continue
            self.ipvs[self.addr[sv.vaddr]] = {"servers": [self.addr[sv.servers[j]] for j in range(sv.n_servers)], "rr_index": 0}
        self.service_addr = [self.addr[w.services[k].vaddr] for k in range(w.struct.n_services)]
        # port 0 = an ephemeral Endpoint: its address is whatever its last bind was given (never a destination operand)
        self.ephemeral = [w.socks[i].port == 0 for i in range(w.struct.n_socks)]
        n_nodes = w.struct.n_nodes
        self.node_ip = {n: (None if w.nodes[n].flags & A.NODE_NO_IP else "10.0.0.%d" % n) for n in range(1, n_nodes + 1)}
        self.addr_to_node = {ip: n for n, ip in self.node_ip.items() if ip is not None}
        self.node_sockets = {n: {} for n in range(0, n_nodes + 1)}        # HashMap<(SocketAddr, protocol), Arc<dyn Socket>>
        # task/mod.rs: TaskHandle.nodes (the current NodeInfo, the paused Runnables, the init fns) and the NodeHandles the test
        # body holds — Spawners over the ORIGINAL NodeInfo (runtime/mod.rs:405-418)
        self.node_info = {n: NodeInfo(n) for n in range(0, n_nodes + 1)}
        self.handle_info = dict(self.node_info)
        self.paused = {n: [] for n in range(0, n_nodes + 1)}
        # NodeBuilder::restart_on_panic_matching: the patterns as STRINGS, and the text of every message code (a literal
        # message the builder interned, else the decimal form of the value `panic!("{}", n)` formats) — the builder's own
        # records, not the 256-bit rows it derived from them for the device and the oracle
        pats, texts = getattr(w, "panic_patterns", {}), getattr(w, "panic_text_of", {})
        self.node_flags = {n: (w.nodes[n].flags, tuple(pats.get(n, ())) if n in pats else
                               tuple(str(w.nodes[n].match[i]) for i in range(min(2, w.nodes[n].n_match))), n in pats)
                           for n in range(0, n_nodes + 1)}
        self.panic_text = lambda code: None if code is None or code == 255 else texts.get(code, str(code))
        self.panic_dyn_max = w.struct.panic_dyn_max or 254
        self.all_socks = []                                                # every EndpointSocket ever made (reset_node's sweep)
        self.hooks_req, self.hooks_rsp = {}, {}                            # NetSim.hooks_req / hooks_rsp: HashMap<NodeId, hook>
        self.cfg, self.rng = cfg, Xoshiro(seed)
        self.clock, self.log, self.logging = 0, [], False
        self.heap, self.ready, self.handles, self.bound = [], [], {}, {}
        self.steps, self.msg_count, self.obs, self.flags = 0, 0, FNV_OFFSET, [0, 0, 0, 0]
        self.clog_in, self.clog_out, self.clog_link = set(), set(), set()
        self.obs_list = []                                                 # what obs folds, in order (debugging aid)
        self.loss = cfg.packet_loss_rate
        self.lat = duration_params(cfg.lat_lo_ns, cfg.lat_hi_ns)
        self.base_ns = 0

    # ---- GlobalRng (rand.rs) ------------------------------------------------------------------------------------
    def with_log(self):
        if self.logging:
            v = (self.rng.peek() >> 32) & 0xFF
            for i in range(8):
                v ^= (self.clock >> (8 * i)) & 0xFF
            self.log.append(v)

    def gen_range(self, lo, hi):                    # rand.with(|r| r.gen_range(lo..hi)): ONE with() per call
        v, _ = gen_range_attempts(self.rng, lo, hi)
        self.with_log()
        return v

    def next_u64(self):                             # RngCore for GlobalRng: one with() per underlying call
        v = self.rng.next()
        self.with_log()
        return v

    def gen_bool(self, p):                          # Bernoulli [DEP A.4]
        if p == 1.0:
            return True
        return self.next_u64() < int(p * 2.0**64)

    def gen_duration_once(self, lo, hi):            # rand.with(|rng| rng.gen_range(lo..hi)) on Durations: ONE with() however many
        mode, low, rg, zone = duration_params(lo, hi)       # attempts the rejection loop takes (task/mod.rs:302-304)
        while True:
            v = self.rng.next()
            if mode == 0:
                m = (v >> 32) * rg
                ok, r = (m & 0xFFFFFFFF) <= zone, low + (m >> 32)
            else:
                m = v * rg
                ok, r = (m & M64) <= zone, low + (m >> 64)
            if ok:
                self.with_log()
                return r

    def sample_duration(self, params):              # UniformDuration::sample on the GlobalRng itself
        mode, low, rg, zone = params
        while True:
            v = self.next_u64()
            if mode == 0:
                m = (v >> 32) * rg
                if (m & 0xFFFFFFFF) <= zone:
                    return low + (m >> 32)
            else:
                m = v * rg
                if (m & M64) <= zone:
                    return low + (m >> 64)

    # ---- Timer: Rust's BinaryHeap with reversed deadline order, by hand --------------------------------------
    def sift_up(self, pos):
        h = self.heap
        hole = h[pos]
        while pos > 0:
            parent = (pos - 1) // 2
            if hole[0] >= h[parent][0]:
                break
            h[pos] = h[parent]; pos = parent
        h[pos] = hole

    def timer_add(self, deadline, cb):
        self.heap.append([deadline, cb]); self.sift_up(len(self.heap) - 1)

    def timer_pop(self):
        h = self.heap
        item = h.pop()
        if h:
            item, h[0] = h[0], item
            end, pos, hole, child = len(h), 0, h[0], 1
            while child + 1 < end:
                if h[child][0] >= h[child + 1][0]:
                    child += 1
                h[pos] = h[child]; pos = child; child = 2 * pos + 1
            if child == end - 1:
                h[pos] = h[child]; pos = child
            h[pos] = hole
            self.sift_up(pos)
        return item

    def expire(self, now):
        while self.heap and self.heap[0][0] <= now:
            _, cb = self.timer_pop()
            self.steps += 1
            cb()

    # ---- tasks (async-task wake rules) --------------------------------------------------------------------------
    def spawn(self, prog, info, record=True):         # Spawner::spawn_inner (:627-654) on the Spawner's NodeInfo
        t = Task(self, prog, info)
        info.tasks.append(t)                        # new_task
        self.ready.append(t)                        # runnable.schedule()
        if record:
            self.handles[prog] = t                  # the JoinHandle the spawning code keeps
        return t

    def spawn_init(self, node):                     # the init closure(s) of a node on its current NodeInfo
        for p in range(1, len(self.progs)):
            if self.progs[p][0] == node and self.progs[p][1] & A.PROG_INIT:
                self.spawn(p, self.node_info[node], record=False)

    # ---- TaskHandle (task/mod.rs:354-424) -----------------------------------------------------------------------
    def paused_clear(self, node):                   # node.paused.clear(): the Runnables drop, their futures with them
        lst, self.paused[node] = self.paused[node], []
        for t in lst:
            self.finish(t, "cancelled")

    def kill(self, node):                           # kill_id
        self.paused_clear(node)
        self.node_info[node].kill(self)
        self.reset_node(node)                       # for sim in sims: sim.reset_node(id)

    def restart(self, node):
        old, self.node_info[node] = self.node_info[node], NodeInfo(node)
        self.paused_clear(node)
        old.kill(self)
        self.spawn_init(node)

    def resume(self, node):
        self.node_info[node].paused = False
        lst, self.paused[node] = self.paused[node], []
        for t in lst:
            self.ready.append(t)                    # sender.send(runnable)

    def reset_node(self, node):                     # Network::reset_node (network.rs:142-147): sockets.clear()
        self.node_sockets[node].clear()
        for sock in self.all_socks:                 # EndpointSockets only the table kept alive die now (table order of the
            if sock["node"] == node and not sock["in_table_or_ep"]():     # entries: the HashMap's own order is not modelled)
                self.sock_free(sock)

    def wake(self, t):
        if not t.alive or t.sched:
            return
        t.sched = True
        if not t.running:
            self.ready.append(t)

    def drop_locals(self, t):                       # the task body's locals: its (tx, rx) first, then its Endpoints (table order)
        self.conn_drop(t)
        for a in sorted(set(t.owned)):
            if self.bound.get(a) is not None and self.bound[a]["owner"] is t:
                self.close_sock(a)

    def drop_guard(self, t):                        # the guard moved into the body drops last: `impl Drop for A { spawn(..) }`
        if self.progs[t.prog][1] & A.PROG_DROP_SPAWN and not t.guard_gone:
            t.guard_gone = True
            self.spawn(t.prog + 1, t.info, record=False)    # Spawner::current(): this task's own Arc<NodeInfo> (task/mod.rs:1185-1253)

    def finish(self, t, outcome):
        t.gen.close()                               # the future is dropped: `finally:` blocks = Drop impls
        self.drop_locals(t)
        self.drop_guard(t)
        t.alive, t.outcome = False, outcome
        if t.joiner is not None:                    # async-task notifies the awaiter
            self.wake(t.joiner)

    # ---- futures ------------------------------------------------------------------------------------------------
    def sleep_deadline(self, deadline):             # TimeHandle::sleep_until: 1 ms floor
        return max(deadline, self.clock + MS)

    def sleep_until(self, t, deadline):             # Sleep::poll
        while self.clock < deadline:
            self.timer_add(deadline, lambda: self.wake(t))
            yield

    def rand_delay(self, t):                        # NetSim::rand_delay (net/mod.rs:287-295)
        delay = self.gen_range(0, 5) * 1000
        if self.cfg.buggify and self.gen_bool(0.1):         # buggify_with_prob(0.1): enabled && with(|rng| rng.gen_bool(p))
            delay = self.gen_range(1, 5) * 10**9
        yield from self.sleep_until(t, self.sleep_deadline(self.clock + delay))

    def close_sock(self, a):                        # drop(Endpoint): conn_rx goes (the async_channel is closed), the guard
        sock = self.bound[a]                        # loses one of its owners; Sender / Receiver clones may keep it alive
        self.bound[a] = None
        sock["conn_closed"], sock["acceptor"], sock["ep_alive"] = True, None, False
        sock["guard"].drop()
        if not sock["in_table_or_ep"]():            # (the table entry went with the guard, or reset_node took it earlier)
            self.sock_free(sock)

    def guard_drop(self, info, node, sock):         # BindGuard::drop (net/mod.rs:483-493)
        if info.killed:                             # "avoid interfering with restarted node"
            return
        if self.node_sockets[node].get(sock["addr"]) is sock:
            del self.node_sockets[node][sock["addr"]]          # Network::close (network.rs:253-258)
        if not sock["in_table_or_ep"]():
            self.sock_free(sock)

    def sock_free(self, sock):
        # the last Arc<EndpointSocket> is gone (table entry and Endpoint; in-flight delivery closures are not counted —
        # DESIGN.md): conn_tx dies, and with it the connections nobody accepted
        q, sock["connq"] = sock["connq"], []
        for tx, rx in q:
            self.tx_drop(tx); self.rx_drop(rx)

    # ---- reliable channel (net/mod.rs:337-430, endpoint.rs:178-262) ---------------------------------------------
    def channel(self, node, dst):                   # NetSim::channel: an mpsc + the test_link closure both ends share
        ch = Mpsc()

        def test_link():
            sent = self.try_send(node, dst)
            return None if sent is None else self.clock + sent[3]
        return (ch, test_link), (ch, test_link)

    def tx_drop(self, tx):                          # PayloadSender dropped: the last mpsc sender -> a parked receiver wakes
        ch = tx[0]
        ch.tx_alive = False
        if ch.rx_task is not None:
            r, ch.rx_task = ch.rx_task, None
            self.wake(r)

    def rx_drop(self, rx):                          # PayloadReceiver dropped: tx.send() fails from now on
        rx[0].rx_alive, rx[0].rx_task = False, None

    def conn_drop(self, t):                         # drop(tx); drop(rx) — each is {_guard, channel end}, fields in that order
        if t.conn is None:
            return
        tx, rx, guard = t.conn
        t.conn = None
        guard.drop(); self.tx_drop(tx)
        guard.drop(); self.rx_drop(rx)

    def connect1(self, t, a, dst):                  # Endpoint::connect1 -> NetSim::connect1
        yield from self.rand_delay(t)
        self.conn_drop(t)                           # (the VM's rule: a task holds one pair; the old one goes here)
        node = self.socks[a][0]
        dst = self.ipvs_get_server(dst) or dst      # net/mod.rs:345-350
        sent = self.try_send(node, dst)
        if sent is None:
            return A.VAL_REFUSED
        src_ip, dst_node, sock, _lat = sent         # "FIXME: delay": the latency is discarded
        src = (src_ip, self.bound[a]["addr"][1])
        tx1, rx1 = self.channel(node, dst)
        tx2, rx2 = self.channel(dst_node, src)
        if sock["conn_closed"]:                     # socket.new_connection: `let _ = conn_tx.try_send(..)`
            self.tx_drop(tx2); self.rx_drop(rx1)
        else:
            sock["connq"].append((tx2, rx1))
            if sock["acceptor"] is not None:
                acc, sock["acceptor"] = sock["acceptor"], None
                self.wake(acc)
        guard = self.bound[a]["guard"]
        t.conn = (tx1, rx2, guard.clone().clone())  # Sender { _guard: guard.clone(), tx } + Receiver { _guard: guard.clone(), rx }
        return 0

    def accept1(self, t, a):                        # Endpoint::accept1
        yield from self.rand_delay(t)
        sock = self.bound[a]
        while not sock["connq"]:                    # conn_rx.recv().await
            sock["acceptor"] = t
            yield
        self.conn_drop(t)
        tx, rx = sock["connq"].pop(0)
        t.conn = (tx, rx, sock["guard"].clone().clone())

    def chan_send(self, t, val):                    # Sender::send -> PayloadSender::send (net/mod.rs:417-421)
        if t.conn is None:
            return A.VAL_RESET
        ch, test_link = t.conn[0]
        state = test_link()                         # the draws come before the closed check
        if not ch.rx_alive:
            return A.VAL_RESET
        ch.q.append((val, state))
        if ch.rx_task is not None:
            r, ch.rx_task = ch.rx_task, None
            self.wake(r)
        return None

    def chan_recv(self, t):                         # Receiver::recv -> the stream of channel() (net/mod.rs:385-402)
        if t.conn is None:
            return A.VAL_RESET
        ch, test_link = t.conn[1]
        while not ch.q:
            if not ch.tx_alive:
                return A.VAL_RESET                  # the stream ended
            ch.rx_task = t
            yield
        val, state = ch.q.pop(0)
        backoff = MS
        while state is None:
            yield from self.sleep_until(t, self.sleep_deadline(self.clock + backoff))
            backoff = min(backoff * 2, 10 * 1000 * MS)
            state = test_link()
        yield from self.sleep_until(t, self.sleep_deadline(state))
        return val

    def ipvs_get_server(self, dst):                 # IpVirtualServer::get_server (net/ipvs.rs:88-105), RoundRobin
        service = self.ipvs.get(dst)
        if service is None or not service["servers"]:
            return None
        if service["rr_index"] >= len(service["servers"]):
            service["rr_index"] = 0
        server = service["servers"][service["rr_index"]]
        service["rr_index"] += 1
        return server

    def resolve_dest_node(self, node, dst):         # network.rs:272-290
        if dst[0] == "127.0.0.1" or dst in self.node_sockets[node]:
            return node
        if self.node_ip[node] is None:
            return None                             # "ip not set"
        return self.addr_to_node.get(dst[0])        # or "destination not found"

    def try_send(self, node, dst):                  # network.rs:296-313 -> (src_ip, socket, latency) or None
        dst_node = self.resolve_dest_node(node, dst)
        if dst_node is None:
            return None
        if node in self.clog_out or dst_node in self.clog_in or (node, dst_node) in self.clog_link:   # test_link (:261-269)
            return None
        if self.gen_bool(self.loss):
            return None
        self.msg_count += 1
        lat = self.sample_duration(self.lat)
        sockets = self.node_sockets[dst_node]
        ep = sockets.get(dst) or sockets.get(("0.0.0.0", dst[1]))
        if ep is None:
            return None
        if dst[0] == "127.0.0.1":
            src_ip = "127.0.0.1"
        else:
            if self.node_ip[node] is None:
                raise Panic()                       # `.ip.unwrap()`
            src_ip = self.node_ip[node]
        return src_ip, dst_node, ep, lat

    def send_raw(self, t, ep, dst, tag, val, aux=0, kind="datagram"):   # Endpoint::send_to_raw -> NetSim::send (net/mod.rs:298-333)
        yield from self.rand_delay(t)
        node = self.socks[ep][0]
        hook = self.hooks_req.get(node)
        if hook is not None and kind == "request" and not hook(tag, val):
            return                                  # `if !hook(&msg) { return Ok(()) }`
        dst = self.ipvs_get_server(dst) or dst      # net/mod.rs:312-317
        sent = self.try_send(node, dst)
        if sent is not None:
            src_ip, dst_node, mbox, lat = sent
            frm = (src_ip, self.addr[ep][1])
            rsp_hook = self.hooks_rsp.get(dst_node)     # cloned now, consulted when the timer fires

            def arrive():
                if rsp_hook is not None and kind == "response" and not rsp_hook(val):
                    return
                self.deliver(mbox, tag, val, frm, aux)
            self.timer_add(self.clock + lat, arrive)

    def mailbox_recv(self, t, ep, tag):             # Mailbox::recv
        mbox, os_ = self.bound[ep], Oneshot()
        os_.rx_task = t
        idx = next((i for i, m in enumerate(mbox["msgs"]) if m[0] == tag), None)
        if idx is not None:
            m = mbox["msgs"][idx]
            mbox["msgs"][idx] = mbox["msgs"][-1]; mbox["msgs"].pop()
            os_.val = m[1:]
        else:
            mbox["regs"].append((tag, os_))
        return os_

    def deliver(self, mbox, tag, val, frm, aux):    # Mailbox::deliver
        regs, i = mbox["regs"], 0
        while i < len(regs):
            if regs[i][0] == tag:
                os_ = regs[i][1]
                regs[i] = regs[-1]; regs.pop()
                if os_.rx_alive:                    # oneshot::Sender::send Ok: value stored, receiver task woken
                    os_.val = (val, frm, aux); self.wake(os_.rx_task)
                    return
            else:
                i += 1
        mbox["msgs"].append((tag, val, frm, aux))

    def recv_raw(self, t, ep, tag):                 # Endpoint::recv_from_raw
        os_ = self.mailbox_recv(t, ep, tag)
        try:
            while os_.val is None:
                yield
            msg = os_.val
        finally:
            os_.rx_alive = False                    # the Receiver is dropped with the future (or consumed)
        yield from self.rand_delay(t)
        return msg

    def timeout(self, t, dur_ns, fut):              # time::timeout: select_biased! { fut, sleep }
        deadline = self.sleep_deadline(self.clock + dur_ns)      # the Sleep exists before the first poll
        try:
            while True:
                try:
                    next(fut)
                except StopIteration as e:
                    return ("ok", e.value)
                if self.clock >= deadline:
                    return ("elapsed", None)
                self.timer_add(deadline, lambda: self.wake(t))   # Sleep::poll: ANOTHER timer on every poll
                yield
        finally:
            fut.close()

    def rpc_call(self, t, ep, dst, req_tag, code):  # Endpoint::call_with_data
        rsp_tag = self.next_u64()                   # random::<u64>()
        yield from self.send_raw(t, ep, dst, req_tag, code, rsp_tag, kind="request")
        val, frm, _ = yield from self.recv_raw(t, ep, rsp_tag)
        if frm != dst:
            raise Panic()
        return val

    # ---- one task body ---------------------------------------------------------------------------------------
    def body(self, t, pc):
        while True:
            op, a, b, imm = self.insns[pc]
            name, dur = OPN[op], b * 10**9 + imm
            nxt = pc + 1
            if name == "DONE":
                if self.progs[t.prog][1] & A.PROG_INIT:   # `async move { future.await; h.exit() }` (runtime/mod.rs:362-370): the
                    self.drop_locals(t)                   # body's locals are gone when `future.await` returns; then Spawner::exit
                    self.drop_guard(t)
                    t.info.kill(self)                     # = info.kill() on the NodeInfo this init task was spawned with
                return
            elif name == "SPAWN":
                # NodeHandle::spawn for another node's program (the handle's ORIGINAL NodeInfo), task::spawn otherwise
                c = self.spawn(a, self.handle_info[self.progs[a][0]] if self.progs[a][0] != t.node else t.info)
                if b & 2:                           # `async move`: the child takes (tx, rx)
                    c.conn, t.conn = t.conn, None
                if b & 4:
                    c.val, c.frm, c.aux = t.val, t.frm, t.aux
            elif name == "JOIN":
                c = self.handles.get(a)
                if c is None:
                    raise Panic()
                while c.alive:
                    c.joiner = t
                    yield
                if (c.outcome == "cancelled") != bool(b & 1):      # .unwrap() / .unwrap_err()
                    raise Panic()
            elif name == "ABORT":                   # AbortHandle::abort (task/join.rs:158-163)
                c = self.handles.get(a)
                if c is not None and c.alive:
                    c.cancelled = True
                    self.wake(c)
            elif name == "KILL":
                self.kill(a)
            elif name == "RESTART":
                self.restart(a)
            elif name == "PAUSE":
                self.node_info[a].paused = True
            elif name == "RESUME":
                self.resume(a)
            elif name == "ASSERT_EXIT":             # Handle::is_exit
                if self.node_info[a].killed != bool(b & 1):
                    raise Panic()
            elif name == "BUILD":                   # create_node().init(..).build() inside a task body
                for p in range(1, len(self.progs)):
                    if self.progs[p][0] == a and (self.progs[p][1] & A.PROG_INIT) and not (self.progs[p][1] & A.PROG_PRE):
                        self.spawn(p, self.node_info[a], record=False)
            elif name == "ADVANCE":                 # time::advance (time/mod.rs:103-106)
                self.clock += dur
                self.expire(self.clock)
            elif name == "YIELD":
                self.wake(t)
                yield
            elif name == "PANIC":
                if a == 0:
                    raise Panic(imm & 0xFF)
                v = (self.flags[b & 3] + imm) & 0xFFFFFFFF      # panic!("{}", flag + imm): values past panic_dyn_max are outside the model
                raise Panic(v if v <= self.panic_dyn_max else 255)
            elif name == "IPVS":                    # NetSim::global_ipvs().{add,del}_{service,server} (net/ipvs.rs:50-85)
                key = self.service_addr[b]
                if a == 0:
                    self.ipvs[key] = {"servers": [], "rr_index": 0}             # HashMap::insert: a fresh Service
                elif a == 1:
                    self.ipvs.pop(key, None)
                else:
                    service = self.ipvs.get(key)
                    if service is None:
                        raise Panic(255)                                         # .expect("service not found")
                    if a == 2:
                        service["servers"].append(self.addr[imm])
                    else:
                        service["servers"] = [x for x in service["servers"] if x != self.addr[imm]]     # retain
            elif name == "HOOK_REQ":                # NetSim::hook_rpc_req::<R>(node, f): HashMap::insert
                self.hooks_req[a] = (lambda tag, code, want_tag=b >> 8, all_=b & 1, want=imm & 0xFF:
                                     not (tag == want_tag and (all_ or code == want)))
            elif name == "HOOK_RSP":
                self.hooks_rsp[a] = (lambda code, all_=b & 1, want=imm & 0xFF: not (all_ or code == want))
            elif name == "SET":
                t.cnt[a & 1] = imm & 0xFFFF
            elif name == "DJNZ":
                t.cnt[a & 1] = (t.cnt[a & 1] - 1) & 0xFFFF
                if t.cnt[a & 1]:
                    nxt = b
            elif name == "JMP":
                nxt = b
            elif name == "JEQ":
                if t.val == imm:
                    nxt = b
            elif name == "TRACE":
                v = imm + (t.cnt[a & 1] if b & 1 else 0)
                self.obs_list.append(v)
                self.obs = ((self.obs ^ v) * FNV_PRIME) & M64
            elif name == "TRACE_TIME":
                v = self.base_ns + self.clock if a == 0 else self.clock if a == 1 else t.val
                self.obs_list.append(v)
                self.obs = ((self.obs ^ v) * FNV_PRIME) & M64
            elif name == "SLEEP":
                yield from self.sleep_until(t, self.sleep_deadline(self.clock + dur))
            elif name == "SLEEP_RAND":
                d = self.sample_duration(duration_params(a * 50 * MS, dur))
                yield from self.sleep_until(t, self.sleep_deadline(self.clock + d))
            elif name == "MARK":
                t.t0 = self.clock
            elif name == "SLEEP_UNTIL":
                yield from self.sleep_until(t, self.sleep_deadline(t.t0 + dur))
            elif name == "ASSERT_ELAPSED":
                el = self.clock - t.t0
                if not {0: el == dur, 1: el >= dur, 2: el < dur}[a]:
                    raise Panic()
            elif name == "BIND":                    # Endpoint::bind -> Network::bind (network.rs:206-251)
                yield from self.rand_delay(t)
                ip, port = self.addr[a]
                err = 0
                # (a table entry is one node's: the workload VM's rule, not the reference's)
                if self.socks[a][0] != t.node:
                    err = A.VAL_ADDR_NOT_AVAILABLE
                elif ip not in ("0.0.0.0", "127.0.0.1") and self.node_ip[t.node] is not None and ip != self.node_ip[t.node]:
                    err = A.VAL_ADDR_NOT_AVAILABLE
                elif self.ephemeral[a]:             # "resolve port if unspecified" (:224-236)
                    port = next((p for p in range(1, 65536) if (ip, p) not in self.node_sockets[t.node]), None)
                    if port is None:
                        err = A.VAL_ADDR_IN_USE
                elif (ip, port) in self.node_sockets[t.node]:
                    err = A.VAL_ADDR_IN_USE
                if err:
                    if not (b & 1):
                        raise Panic()               # .unwrap()
                    t.val = err
                else:
                    mbox = dict(owner=t, regs=[], msgs=[], addr=(ip, port), connq=[], conn_closed=False, acceptor=None,
                                node=t.node, ep_alive=True)
                    mbox["in_table_or_ep"] = (lambda sock=mbox, node=t.node:
                                              sock["ep_alive"] or self.node_sockets[node].get(sock["addr"]) is sock)
                    mbox["guard"] = Arc(lambda info=t.info, node=t.node, sock=mbox: self.guard_drop(info, node, sock))
                    self.all_socks.append(mbox)
                    self.node_sockets[t.node][(ip, port)] = mbox
                    self.bound[a] = mbox; t.owned.append(a)
                    self.addr[a] = (ip, port)       # ep.local_addr()
                    if b & 1:
                        t.val = 0
                    if b & 2:
                        t.val = port
            elif name == "CLOSE":
                if self.bound.get(a) is not None and self.bound[a]["owner"] is t:
                    self.close_sock(a)
            elif name == "CONNECT":
                t.val = yield from self.connect1(t, a, self.addr[b & 0xFF])
            elif name == "ACCEPT":
                yield from self.accept1(t, a)
            elif name == "CSEND":
                r = self.chan_send(t, imm)
                if r is not None:
                    t.val = r
            elif name == "CRECV":
                t.val = yield from self.chan_recv(t)
            elif name == "CCLOSE":
                self.conn_drop(t)
            elif name == "CLOG_LINK":
                self.clog_link.add((a, b))
            elif name == "UNCLOG_LINK":
                self.clog_link.discard((a, b))
            elif name == "SEND":
                yield from self.send_raw(t, a, self.addr[b & 0xFF], b >> 8, imm)
            elif name == "REPLY":
                yield from self.send_raw(t, a, t.frm, b >> 8, imm)
            elif name == "RPC_REPLY":
                yield from self.send_raw(t, a, t.frm, t.aux, imm & 0xFF, kind="response")
            elif name == "RECV":
                val, frm, aux = yield from self.recv_raw(t, a, b >> 8)
                t.val, t.frm = val, frm
                if (b >> 8) >= 0x80:
                    t.aux = aux
            elif name == "RECV_TIMEOUT":
                how, msg = yield from self.timeout(t, (b & 0xFF) * 10**9 + imm, self.recv_raw(t, a, b >> 8))
                if how == "ok":
                    t.val, t.frm = msg[0], msg[1]
                    if (b >> 8) >= 0x80:
                        t.aux = msg[2]
                else:
                    t.val = A.VAL_TIMEOUT
            elif name == "RPC_CALL":
                call = self.rpc_call(t, a, self.addr[b & 0xFF], b >> 8, imm & 0xFF)
                if imm >> 8:
                    how, v = yield from self.timeout(t, (imm >> 8) * MS, call)
                    t.val = v if how == "ok" else A.VAL_TIMEOUT
                    if how == "ok":
                        t.frm = self.addr[b & 0xFF]
                else:
                    t.val = yield from call
                    t.frm = self.addr[b & 0xFF]
            elif name == "ASSERT_VAL":
                if t.val != imm:
                    raise Panic()
            elif name == "RAND_BOOL":
                t.val = 1 if self.gen_bool(self.cfg.loss_table[a & 3]) else 0
            elif name == "RANDOM":
                v = self.next_u64()
                t.val = (v >> 32) if a == 0 else (v >> 32) & 0xFF
            elif name == "GSET":
                self.flags[a & 3] = imm
            elif name == "GADD":
                self.flags[a & 3] = (self.flags[a & 3] + imm) & 0xFFFFFFFF
            elif name == "ASSERT_G":
                if self.flags[a & 3] != imm:
                    raise Panic()
            elif name == "PANIC_IF_G_LT":
                if self.flags[a & 3] < imm:
                    raise Panic()
            elif name == "CLOG_NODE":
                if b & 1: self.clog_in.add(a)
                if b & 2: self.clog_out.add(a)
            elif name == "UNCLOG_NODE":
                if b & 1: self.clog_in.discard(a)
                if b & 2: self.clog_out.discard(a)
            elif name == "SET_LOSS":
                self.loss = self.cfg.loss_table[a & 3]
            elif name == "SET_LATENCY":                      # NetSim::update_config(|c| c.send_latency = ..) (net/mod.rs:138-141)
                self.lat = duration_params(self.cfg.lat_table_lo_ns[a & 3], self.cfg.lat_table_hi_ns[a & 3])
            else:
                raise NotImplementedError(name)
            pc = nxt

    # ---- Executor::block_on ---------------------------------------------------------------------------------------
    def run(self, time_limit=0):
        self.base_ns = (60 * 60 * 24 * 365 * 52 + self.gen_range(0, 60 * 60 * 24 * 365)) * 10**9   # not logged
        self.logging = True
        for p in range(1, len(self.progs)):         # `node.spawn(..)` / init tasks of nodes built ahead of block_on, in order
            if self.progs[p][1] & A.PROG_PRE:
                self.spawn(p, self.node_info[self.progs[p][0]], record=not (self.progs[p][1] & A.PROG_INIT))
        main = self.spawn(0, self.node_info[0], record=False)
        verdict = A.PASS
        while True:
            panicked = False
            while self.ready:
                idx = self.gen_range(0, len(self.ready))
                t = self.ready[idx]
                self.ready[idx] = self.ready[-1]; self.ready.pop()
                if t.cancelled or t.info.killed:    # cancelled task or killed node: drop the future (:269-273)
                    self.steps += 1
                    self.finish(t, "cancelled")
                elif t.info.paused:                 # paused task: push to the node's waiting list, no poll, no time advance
                    self.paused[t.info.node].append(t)
                    continue
                else:
                    self.steps += 1
                    t.sched, t.running = False, True
                    try:
                        next(t.gen)
                    except StopIteration:
                        self.finish(t, "completed")
                    except Panic as e:
                        flags, patterns, substr = self.node_flags[t.info.node]
                        error_msg = self.panic_text(e.code)         # None: a failed assert / unwrap — a message no pattern names
                        # restart_on_panic || restart_on_panic_matching.iter().any(|s| error_msg.contains(s))  (task/mod.rs:297-300);
                        # raw C-ABI tables without the builder's records carry codes, compared for equality
                        hit = error_msg is not None and any((p in error_msg) if substr else (p == error_msg) for p in patterns)
                        if not (flags & A.NODE_RESTART_ON_PANIC or (flags & A.NODE_RESTART_MATCHING and hit)):
                            panicked = True         # resume_unwind
                            break
                        # async-task's guard drops the future while unwinding (the node is not killed yet), then (:301-313)
                        self.finish(t, "cancelled")
                        delay = self.gen_duration_once(1 * 10**9, 10 * 10**9)
                        node = t.info.node
                        self.kill(node)
                        self.timer_add(self.clock + delay, lambda node=node: self.restart(node))
                    if t.alive:
                        t.running = False
                        if t.sched:
                            self.ready.append(t)
                self.clock += self.gen_range(50, 100)
                self.expire(self.clock)
            if panicked:
                verdict = A.PANIC; break
            if not main.alive:
                break
            if not self.heap:
                verdict = A.DEADLOCK; break
            now = self.heap[0][0] + 50
            self.expire(now)
            self.clock = now
            if time_limit and self.clock >= time_limit:
                verdict = A.TIME_LIMIT; break
        h = FNV_OFFSET
        for v in self.log:
            h = ((h ^ v) * FNV_PRIME) & M64
        return dict(verdict=verdict, steps=self.steps, clock_ns=self.clock, msg_count=self.msg_count,
                    rng_calls=self.rng.calls, trace_hash=h, obs_hash=self.obs, log=bytes(self.log).hex())


def workloads():
    """name -> (workload, the Config it was generated for or None).  Every entry runs under Config::default() and 20 % loss;
    entries with a Config of their own (non-default latency ranges, buggify, loss tables) under that one too."""
    import random
    from tests import fuzz
    from tests import lifecycle_workloads as LW
    out = {"pingpong_4x2": (W.pingpong(4, 2), None)}
    # every reference-shaped workload of tests/lifecycle_workloads.py: timeouts, dropped receivers, typed RPC, hooks, address
    # resolution, ephemeral ports, the reliable channel, and the node lifecycle (kill / restart / pause / resume / abort /
    # init + exit / restart_on_panic[_matching]) restated literally in this file
    for name in sorted(LW.ALL):
        out[name] = (LW.ALL[name](), LW.config(name))
    # a lossy, clogged RPC retry loop: call_timeout until it succeeds, while the supervisor clogs the server for a while
    wl = W.WorkloadBuilder()
    ns, nc = wl.create_node(), wl.create_node()
    asv, acl = wl.addr(ns, 1), wl.addr(nc, 1)
    h = wl.task(ns); h.sleep(ms=3); h.rpc_reply(asv, 9)
    s = wl.task(ns); s.bind(asv)
    top = s.label(); s.rpc_recv(asv, 2); s.spawn(h, move_request=True); s.jmp(top)
    c = wl.task(nc); c.bind(acl); c.sleep(ms=5); c.set(0, 6)
    top = c.label(); c.rpc_call(acl, asv, 2, 77, timeout_ms=40); c.trace_val(); c.sleep_rand(lo_ms=0, ms=30); c.djnz(0, top)
    m = wl.main(); m.spawn(s); m.spawn(c); m.sleep(ms=60); m.clog_node(ns, "in"); m.sleep(ms=90); m.unclog_node(ns, "in"); m.join(c)
    out["rpc_retry_under_clog"] = (wl.build(), None)
    # random programs of every generator in tests/fuzz.py (fixed seeds)
    for gen, base, n in (("random_workload", 810000, 16), ("random_lifecycle_workload", 820000, 24), ("random_rpc_workload", 830000, 16),
                         ("random_addr_workload", 880000, 24), ("random_ephemeral_workload", 870000, 16),
                         ("random_channel_workload", 860000, 24), ("random_guard_workload", 850000, 16),
                         ("random_supervisor_workload", 840000, 16), ("random_mixed_workload", 845000, 16),
                         ("random_ipvs_workload", 895000, 16), ("random_ipvs_runtime_workload", 897000, 16),
                         ("random_latency_workload", 898000, 24)):
        for k in range(n):
            r = getattr(fuzz, gen)(random.Random(base + k))
            out["%s_%02d" % (gen.replace("random_", "fuzz_").replace("_workload", ""), k)] = (r[0], r[1])
    for k in range(12):                             # random typed-RPC programs with hooks
        r = fuzz.random_rpc_workload(random.Random(890000 + k), hooks=True)
        out["fuzz_rpc_hooks_%02d" % k] = (r[0], r[1])
    return out


CFG_FIELDS = ("packet_loss_rate", "lat_lo_ns", "lat_hi_ns", "buggify")


def cfg_to_json(cfg):
    d = dict({f: getattr(cfg, f) for f in CFG_FIELDS}, loss_table=[cfg.loss_table[i] for i in range(4)])
    if cfg.n_lat_table:                                 # (only where used: the entries of rounds 1-5 keep their bytes)
        d["lat_table"] = [[cfg.lat_table_lo_ns[i], cfg.lat_table_hi_ns[i]] for i in range(cfg.n_lat_table)]
    return d


def cfg_from_json(d):
    return A.Config.default(packet_loss_rate=d["packet_loss_rate"], lat_lo_ns=d["lat_lo_ns"], lat_hi_ns=d["lat_hi_ns"],
                            buggify=bool(d["buggify"]), loss_table=tuple(d["loss_table"]),
                            lat_table=tuple(tuple(r) for r in d.get("lat_table", ())))


def main():
    import hashlib
    ex = {}
    for name, (w, own) in workloads().items():
        ex[name] = {}
        # (a workload that switches latencies needs its table under every config: an op naming a missing entry is refused)
        tab = tuple((own.lat_table_lo_ns[i], own.lat_table_hi_ns[i]) for i in range(own.n_lat_table)) if own is not None else ()
        cfgs = [("default", A.Config.default(lat_table=tab)), ("loss20", A.Config.default(packet_loss_rate=0.2, lat_table=tab))]
        if own is not None:
            cfgs.append(("own", own))
            ex[name]["_own_config"] = cfg_to_json(own)
        fuzzed = name.startswith("fuzz_")
        for cfgname, cfg in cfgs:
            ex[name][cfgname] = {}
            for seed in ((0, 1, 99) if fuzzed else (0, 1, 2, 3, 99, 123456789)):
                r = Sim(w, cfg, seed).run()
                if fuzzed:                          # random programs: the log's digest instead of its bytes
                    r["log_sha256"] = hashlib.sha256(bytes.fromhex(r.pop("log"))).hexdigest()[:32]
                ex[name][cfgname][str(seed)] = r
    json.dump(ex, open(os.path.join(HERE, "executor_kat_async.json"), "w"), indent=0, separators=(",", ":"))
    print("wrote executor_kat_async.json", len(ex), "workloads")


if __name__ == "__main__":
    main()
