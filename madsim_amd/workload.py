"""Workload assembler: a madsim test body written as an actor program.

A GPU lane cannot poll an opaque Rust future, so the body of a `#[madsim::test]`
is handed to the runner as a table of instructions, one per reference API call.
The method names below are the reference's own (Endpoint::bind / send_to /
recv_from in net/endpoint.rs, time::sleep in time/sleep.rs, spawn / JoinHandle
in task/mod.rs + task/join.rs, Handle::kill/restart/pause/resume in
runtime/mod.rs:276-303, NetSim::clog_* in net/mod.rs:164-222), so a workload
reads like the Rust test it stands for.

Pure host-side table construction: no device code, no oracle.
"""
import ctypes as C

from . import _abi as A

NS_PER_S = 1_000_000_000
PING, PONG = 0x676E6970, 0x676E6F70  # b"ping", b"pong" little-endian


def _dur(secs=0, ms=0, us=0, ns=0):
    total = int(secs * NS_PER_S) + ms * 1_000_000 + us * 1_000 + ns
    b, imm = divmod(total, NS_PER_S)
    if b > 0xFFFF:
        raise ValueError("duration too long for the 16-bit seconds field")
    return b, imm


class TaskBuilder:
    """The async block of one task (`node.spawn(async move { ... })`)."""

    def __init__(self, wl, index, node, flags=0):
        self.wl, self.index, self.node, self.flags = wl, index, node, flags
        self.code = []  # (op, a, b, imm, reloc)

    def _emit(self, op, a=0, b=0, imm=0, reloc=False):
        self.code.append([A.OP[op], a, b, imm & 0xFFFFFFFF, reloc])
        return self

    # -- control ---------------------------------------------------------------------------------
    def label(self):
        return len(self.code)

    def done(self):
        return self._emit("DONE")

    def spawn(self, task, move_conn=False, move_request=False):
        """`move_conn`: the child's `async move` block takes this task's (tx, rx); `move_request`: it takes the
        request just received on a typed tag (value, sender, response tag) — the per-request task of rpc.rs:170."""
        return self._emit("SPAWN", a=task.index, b=(2 if move_conn else 0) | (4 if move_request else 0))

    def join(self, task, expect_err=False):
        return self._emit("JOIN", a=task.index, b=1 if expect_err else 0)

    def abort(self, task):
        return self._emit("ABORT", a=task.index)

    def yield_now(self):
        return self._emit("YIELD")

    def panic(self, code=0):
        """panic!() with message code `code` (0..254), the thing restart_on_panic_matching patterns name — or panic!("<text>")
        with a literal message: WorkloadBuilder.build() then gives every message the code of its class under the nodes'
        string patterns (`error_msg.contains(pattern)`, task/mod.rs:297-300)."""
        if isinstance(code, str):
            self.wl.panic_texts.append(code)
            self.code.append([A.OP["PANIC"], 0, 0, ("panic_text", code), False])      # the code is assigned in build()
            return self
        return self._emit("PANIC", a=0, imm=code)

    def panic_with_flag(self, flag, offset=0):
        """panic!("{}", flag.load() + offset): the message code is the flag's value."""
        return self._emit("PANIC", a=1, b=flag, imm=offset & 0xFFFFFFFF)

    def set(self, reg, value):
        return self._emit("SET", a=reg, imm=value)

    def djnz(self, reg, target):
        return self._emit("DJNZ", a=reg, b=target, reloc=True)

    def jmp(self, target):
        return self._emit("JMP", b=target, reloc=True)

    def trace(self, value, add_reg=None):
        return self._emit("TRACE", a=add_reg or 0, b=1 if add_reg is not None else 0, imm=value)

    def build_node(self, node):
        return self._emit("BUILD", a=node)

    # -- time ------------------------------------------------------------------------------------
    def sleep(self, **kw):
        b, imm = _dur(**kw)
        return self._emit("SLEEP", b=b, imm=imm)

    def mark(self):
        return self._emit("MARK")

    def sleep_until(self, **kw):
        b, imm = _dur(**kw)
        return self._emit("SLEEP_UNTIL", b=b, imm=imm)

    def assert_elapsed(self, cmp, **kw):
        b, imm = _dur(**kw)
        return self._emit("ASSERT_ELAPSED", a={"==": 0, ">=": 1, "<": 2}[cmp], b=b, imm=imm)

    def advance(self, **kw):
        b, imm = _dur(**kw)
        return self._emit("ADVANCE", b=b, imm=imm)

    # -- net -------------------------------------------------------------------------------------
    def try_bind(self, addr):
        """val = match Endpoint::bind(addr).await { Ok(_) => 0, Err(e) => VAL_ADDR_NOT_AVAILABLE / VAL_ADDR_IN_USE }"""
        return self._emit("BIND", a=addr, b=1)

    def bind(self, addr, port_to_val=False):
        """Endpoint::bind(addr).await.unwrap(); port_to_val: val = ep.local_addr().unwrap().port() (what an ephemeral bind got)"""
        return self._emit("BIND", a=addr, b=2 if port_to_val else 0)

    def send_to(self, ep, dst, tag, val):
        return self._emit("SEND", a=ep, b=(tag << 8) | dst, imm=val)

    def reply(self, ep, tag, val):
        return self._emit("REPLY", a=ep, b=tag << 8, imm=val)

    def recv_from(self, ep, tag):
        return self._emit("RECV", a=ep, b=tag << 8)

    def recv_from_timeout(self, ep, tag, **kw):
        b, imm = _dur(**kw)
        if b > 0xFF:
            raise ValueError("timeout too long")
        return self._emit("RECV_TIMEOUT", a=ep, b=(tag << 8) | b, imm=imm)

    def assert_val(self, val):
        return self._emit("ASSERT_VAL", imm=val)

    # -- typed RPC (Endpoint::call / call_timeout / add_rpc_handler, net/rpc.rs:96-180) -------------
    def rpc_call(self, ep, dst, req_id, code, timeout_ms=0):
        """val = ep.call(dst, R{code}).await, or call_timeout(.., timeout_ms) -> VAL_TIMEOUT.  `req_id` is R::ID - 0x80."""
        if not (0 <= code <= 0xFF and 0 <= timeout_ms < (1 << 24) and 0 <= req_id <= 0x7D):
            raise ValueError("rpc_call: code is 8 bits, timeout_ms 24 bits, req_id 0..125")
        return self._emit("RPC_CALL", a=ep, b=((0x80 + req_id) << 8) | dst, imm=(timeout_ms << 8) | code)

    def hook_rpc_req(self, node, req_id, code=None):
        """NetSim::current().hook_rpc_req::<R>(node, |req| ..): drop requests R sent from `node` — those with request code
        `code`, or all of them when code is None (net/mod.rs:240-262)."""
        return self._emit("HOOK_REQ", a=node, b=((0x80 + req_id) << 8) | (1 if code is None else 0), imm=code or 0)

    def hook_rpc_rsp(self, node, code=None):
        """NetSim::current().hook_rpc_rsp::<R>(node, |rsp| ..): drop responses on their way to `node` (net/mod.rs:264-284)."""
        return self._emit("HOOK_RSP", a=node, b=1 if code is None else 0, imm=code or 0)

    # -- IP Virtual Server at run time (NetSim::global_ipvs(), net/ipvs.rs:50-85) ---------------------
    def ipvs_add_service(self, service):
        """ipvs.add_service(addr, RoundRobin): HashMap::insert of a fresh Service — no servers, rr_index 0, also when it exists."""
        return self._emit("IPVS", a=A.IPVS_ADD_SERVICE, b=service)

    def ipvs_del_service(self, service):
        """ipvs.del_service(addr): the service and its servers are gone; sends to its address are no longer rewritten."""
        return self._emit("IPVS", a=A.IPVS_DEL_SERVICE, b=service)

    def ipvs_add_server(self, service, server):
        """ipvs.add_server(addr, server): servers.push; panics ("service not found") when the service is absent."""
        return self._emit("IPVS", a=A.IPVS_ADD_SERVER, b=service, imm=server)

    def ipvs_del_server(self, service, server):
        """ipvs.del_server(addr, server): servers.retain(|a| a != server); rr_index stays (get_server wraps it, ipvs.rs:96-98)."""
        return self._emit("IPVS", a=A.IPVS_DEL_SERVER, b=service, imm=server)

    def rpc_recv(self, ep, req_id):
        """(req, from) = recv_from_raw(R::ID) of a handler loop (rpc.rs:161): val = request code."""
        return self._emit("RECV", a=ep, b=(0x80 + req_id) << 8)

    def rpc_reply(self, ep, code):
        """send_to_raw(from, rsp_tag, rsp) for the request in hand (rpc.rs:172-175)."""
        return self._emit("RPC_REPLY", a=ep, imm=code & 0xFF)

    # -- reliable channel (Endpoint::connect1 / accept1, net/mod.rs:337-430) -----------------------
    def connect1(self, ep, dst):
        return self._emit("CONNECT", a=ep, b=dst)

    def accept1(self, ep):
        return self._emit("ACCEPT", a=ep)

    def chan_send(self, val):
        return self._emit("CSEND", imm=val)

    def chan_recv(self):
        return self._emit("CRECV")

    def chan_close(self):
        return self._emit("CCLOSE")

    def close(self, ep):
        return self._emit("CLOSE", a=ep)

    # -- supervisor ------------------------------------------------------------------------------
    def kill(self, node):
        return self._emit("KILL", a=node)

    def restart(self, node):
        return self._emit("RESTART", a=node)

    def pause(self, node):
        return self._emit("PAUSE", a=node)

    def resume(self, node):
        return self._emit("RESUME", a=node)

    def clog_node(self, node, direction="both"):
        return self._emit("CLOG_NODE", a=node, b={"in": 1, "out": 2, "both": 3}[direction])

    def unclog_node(self, node, direction="both"):
        return self._emit("UNCLOG_NODE", a=node, b={"in": 1, "out": 2, "both": 3}[direction])

    def clog_link(self, src, dst):
        return self._emit("CLOG_LINK", a=src, b=dst)

    def unclog_link(self, src, dst):
        return self._emit("UNCLOG_LINK", a=src, b=dst)

    def assert_exit(self, node, expected):
        return self._emit("ASSERT_EXIT", a=node, b=1 if expected else 0)

    def set_loss(self, table_index):
        return self._emit("SET_LOSS", a=table_index)

    def set_latency(self, table_index):
        """NetSim::current().update_config(|c| c.send_latency = lat_table[table_index]) (net/mod.rs:138-141)"""
        return self._emit("SET_LATENCY", a=table_index)

    # -- shared flags (Arc<AtomicUsize> in the reference's tests) ------------------------------------
    def flag_store(self, flag, value):
        return self._emit("GSET", a=flag, imm=value)

    def flag_add(self, flag, value):
        return self._emit("GADD", a=flag, imm=value)

    def assert_flag(self, flag, value):
        return self._emit("ASSERT_G", a=flag, imm=value)

    def panic_if_flag_lt(self, flag, value):
        return self._emit("PANIC_IF_G_LT", a=flag, imm=value)

    def sleep_rand(self, lo_ms=0, **kw):
        """sleep(thread_rng().gen_range(lo..hi)): lo in multiples of 50 ms, hi as keyword duration."""
        if lo_ms % 50 or lo_ms > 255 * 50:
            raise ValueError("lo_ms must be a multiple of 50 ms up to 12750 ms")
        b, imm = _dur(**kw)
        return self._emit("SLEEP_RAND", a=lo_ms // 50, b=b, imm=imm)

    def random_u32(self):
        """val = thread_rng().gen::<u32>()"""
        return self._emit("RANDOM", a=0)

    def getrandom_byte(self):
        """val = the byte of getrandom(&mut [0u8; 1]) (the libc interposer routes it to the GlobalRng, rand.rs:197-211)"""
        return self._emit("RANDOM", a=1)

    def trace_system_time(self):
        """observe SystemTime::now() (per-seed base time around 2022 + elapsed)"""
        return self._emit("TRACE_TIME", a=0)

    def trace_val(self):
        """observe the value last received / drawn"""
        return self._emit("TRACE_TIME", a=2)

    def trace_instant(self):
        """observe Instant::now() relative to the runtime's start"""
        return self._emit("TRACE_TIME", a=1)

    def rand_bool(self, table_index):
        """val = thread_rng().gen_bool(config.loss_table[table_index]) as u32 (one draw unless p == 1)."""
        return self._emit("RAND_BOOL", a=table_index)

    def jeq(self, value, target):
        return self._emit("JEQ", b=target, imm=value, reloc=True)


class BuiltWorkload:
    """Owns the ctypes arrays a `madsim_workload_t` points into."""

    def __init__(self, nodes, progs, socks, insns, services=(), panic_match=None, panic_dyn_max=0):
        self.nodes = (A.Node * len(nodes))(*nodes)
        self.progs = (A.Prog * len(progs))(*progs)
        self.socks = (A.Sock * max(1, len(socks)))(*socks)
        self.insns = (A.Insn * len(insns))(*insns)
        self.services = (A.Service * max(1, len(services)))(*services)
        # rows of 8 words per node: bit c = "a panic with message code c restarts the node" (madsim_workload_t.panic_match)
        self.panic_match = (C.c_uint32 * (8 * len(nodes)))(*panic_match) if panic_match is not None else None
        self.struct = A.Workload(len(nodes) - 1, len(progs), len(socks), len(insns), self.nodes, self.progs,
                                 self.socks, self.insns, len(services), panic_dyn_max,
                                 self.services if services else None,
                                 self.panic_match if panic_match is not None else None)

    def ref(self):
        return C.byref(self.struct)


PAYLOAD_BASE = 0x40000000


class WorkloadBuilder:
    def __init__(self):
        self.nodes = [A.Node()]  # node 0 = "madsim-main"
        self.socks = []
        self.tasks = [TaskBuilder(self, 0, 0)]
        self.payloads = []       # interned byte strings: see payload()
        self.rpc_messages = []   # interned typed-RPC (message, data) pairs: see rpc_message()
        self.panic_texts = []    # literal panic messages (TaskBuilder.panic(".."))
        self.panic_patterns = {} # node -> its restart_on_panic_matching patterns, as strings
        self.services = []       # IPVS virtual services: see ipvs_service()

    def payload(self, data: bytes):
        """The 32-bit value standing for the byte string `data` on the wire.  Payload bytes never steer the simulation
        (a Payload is a Box<dyn Any> the network moves around, endpoint.rs:69-94): only the test body looks at them, and all
        it can do in this VM is compare.  So byte strings are interned here — equal bytes <=> equal value — and
        `assert_eq!(&buf[..len], b"..")` becomes assert_val(payload(b"..")).  Four-byte strings stand for themselves
        (little-endian, like PING / PONG) unless they would look like an id or an error value; every other string gets
        PAYLOAD_BASE + its index in the pool.  BuiltWorkload.payloads maps back."""
        data = bytes(data)
        v = int.from_bytes(data, "little")
        if len(data) == 4 and not PAYLOAD_BASE <= v < PAYLOAD_BASE + 0x10000 and v < 0xFFFFFF00:
            return v
        if data not in self.payloads:
            self.payloads.append(data)
        return PAYLOAD_BASE + self.payloads.index(data)

    def rpc_message(self, msg, data: bytes = b""):
        """The 8-bit code standing for a typed-RPC message on the wire: the request (or response) value `msg` together with the
        bytes of `call_with_data` / `add_rpc_handler_with_data` (net/rpc.rs:114-131,143-180).  Like payload(): neither the
        value nor the bytes steer the simulation, the test body only compares them — so (msg, data) pairs are interned, equal
        pairs <=> equal codes, at most 256 distinct pairs per workload.  Use the code as `code` of rpc_call / rpc_reply /
        hook_rpc_*, and assert_val(rpc_message(..)) for `assert_eq!((rsp, &data[..]), (.., b".."))`.
        BuiltWorkload.rpc_messages maps back."""
        key = (msg, bytes(data))
        if key not in self.rpc_messages:
            if len(self.rpc_messages) >= 256:
                raise ValueError("at most 256 distinct typed-RPC (message, data) pairs per workload")
            self.rpc_messages.append(key)
        return self.rpc_messages.index(key)

    def received(self, data: bytes, buf_len: int):
        """What `recv_from(tag, &mut buf)` with a `buf_len`-byte buffer leaves of a message `data` (endpoint.rs:87-94):
        (the value of &buf[..len], len) with len = min(buf.len(), data.len()) — a host-side computation, like the copy."""
        n = min(buf_len, len(data))
        return self.payload(bytes(data)[:n]), n

    def main(self):
        """The future handed to Runtime::block_on / the #[madsim::test] body."""
        return self.tasks[0]

    def create_node(self, restart_on_panic=False, restart_on_panic_matching=(), ip=True):
        """create_node()[.ip(10.0.0.<id>)][.restart_on_panic()][.restart_on_panic_matching(code)...] (runtime/mod.rs:377-395).
        ip=False: the node has no address — it can bind anything, but cannot reach other nodes (network.rs:281-283)."""
        n = A.Node()
        n.flags = (A.NODE_RESTART_ON_PANIC if restart_on_panic else 0) | (0 if ip else A.NODE_NO_IP)
        if restart_on_panic_matching:
            # Patterns are substrings of the panic message (`error_msg.contains(s)`, task/mod.rs:297-300).  A number stands for
            # its decimal text — the reference's own test panics with `panic!("{}", n)` and matches "0" and "1" — so any
            # number of patterns, strings and numbers alike; build() evaluates them against every message code.
            pats = tuple(p if isinstance(p, str) else str(int(p)) for p in restart_on_panic_matching)
            if any(p == "" for p in pats):
                raise ValueError("an empty pattern matches every message: use restart_on_panic")
            self.panic_patterns[len(self.nodes)] = pats
            n.flags |= A.NODE_RESTART_MATCHING
        self.nodes.append(n)
        return len(self.nodes) - 1

    def addr(self, node, port, ip="node"):
        """A SocketAddr an Endpoint may bind or send to: 10.0.0.<node>:<port> (ip="node"), or — as used on `node` —
        0.0.0.0:<port> (ip="unspecified") / 127.0.0.1:<port> (ip="loopback").  Resolution as in network.rs:272-313.
        port=0 is an ephemeral Endpoint (`Endpoint::bind("0.0.0.0:0")`): each bind takes the lowest port from 1 up that no
        socket of the node holds for that IP (network.rs:224-236).  Such an entry cannot be a destination operand: peers
        reply to `from`, or dial a named entry that carries the port it was given."""
        kind = {"node": A.ADDR_IP, "unspecified": A.ADDR_UNSPECIFIED, "loopback": A.ADDR_LOOPBACK}[ip]
        if not 0 <= port <= 0xFFFF:
            raise ValueError("port out of range")
        self.socks.append(A.Sock(node, kind, port))
        return len(self.socks) - 1

    def virtual_addr(self, ip_id, port):
        """A virtual service address that belongs to no node ("1.1.1.<ip_id>:<port>"): a destination only.  Without an IPVS
        service a datagram sent there is dropped before any draw ("destination not found", network.rs:285-289)."""
        if not 1 <= ip_id <= 255 or not 1 <= port <= 0xFFFF:
            raise ValueError("virtual address: ip id 1..255, port 1..65535")
        self.socks.append(A.Sock(ip_id, A.ADDR_VIRTUAL, port))
        return len(self.socks) - 1

    def ipvs_service(self, vaddr, servers=(), absent=False):
        """`ipvs.add_service(ServiceAddr::Tcp(vaddr), Scheduler::RoundRobin)` + one `add_server` per entry of `servers`, before
        any task runs (net/ipvs.rs:50-85; the reference test net/tcp/mod.rs:254-315): send_to / call / connect1 towards `vaddr`
        go to the servers in turn (net/mod.rs:312-317,345-350).  absent=True only declares the address: the service does not
        exist until a task calls ipvs_add_service.  Tasks change services at run time with ipvs_add_service / ipvs_del_service /
        ipvs_add_server / ipvs_del_server."""
        servers = list(servers)
        if len(self.services) >= A.MAX_SERVICES or len(servers) > 6 or (absent and servers):
            raise ValueError("at most 8 services with at most 6 servers each; a service declared absent has none")
        sv = A.Service(vaddr, A.SERVICE_ABSENT if absent else len(servers))
        for i, e in enumerate(servers):
            sv.servers[i] = e
        self.services.append(sv)
        return len(self.services) - 1

    def task(self, node, init=False, pre=False, spawn_on_drop=False):
        """A task body.  spawn_on_drop: the body owns a guard whose Drop calls task::spawn(<the NEXT task declared>) — it
        runs whenever an instance returns or is dropped (abort, kill, panic), in that instance's context (task/mod.rs:1185-1253)."""
        t = TaskBuilder(self, len(self.tasks), node,
                        (A.PROG_INIT if init else 0) | (A.PROG_PRE if pre else 0) | (A.PROG_DROP_SPAWN if spawn_on_drop else 0))
        self.tasks.append(t)
        return t

    def _panic_rows(self):
        """Message codes and the nodes' restart rows.  Every message a task can panic with gets an 8-bit code: a run-time
        formatted `panic!("{}", n)` (panic_with_flag) and a numeric panic(code) ARE their value, whose text is its decimal
        form; literal messages are interned from 254 downwards, and formatted values may not reach into that range
        (panic_dyn_max: the device answers MADSIM_OVERFLOW beyond it).  A node's row has bit c set when one of its patterns
        is a substring of the text of code c — `restart_on_panic_matching.iter().any(|s| error_msg.contains(s))`
        (task/mod.rs:297-300) evaluated once per code on the host."""
        texts = sorted(set(self.panic_texts))
        if len(texts) > 200:
            raise ValueError("too many distinct literal panic messages (at most 200)")
        codes = {m: 254 - i for i, m in enumerate(texts)}
        dyn_max = 254 - len(texts)
        text_of = {c: str(c) for c in range(dyn_max + 1)}
        text_of.update({c: m for m, c in codes.items()})
        rows = [0] * (8 * len(self.nodes))
        for n, pats in self.panic_patterns.items():
            for c, text in text_of.items():
                if any(p in text for p in pats):
                    rows[8 * n + (c >> 5)] |= 1 << (c & 31)
        return codes, rows, dyn_max

    def build(self):
        insns, progs = [], []
        rows, dyn_max = None, 0
        if self.panic_texts or self.panic_patterns:
            codes, rows, dyn_max = self._panic_rows()
            for t in self.tasks:
                for op, a, b, imm, reloc in t.code:
                    if op == A.OP["PANIC"] and a == 0 and not isinstance(imm, tuple) and imm > dyn_max:
                        raise ValueError(f"panic({imm}): numeric message codes above {dyn_max} are taken by this workload's literal messages")
                t.code = [(op, a, b, codes[imm[1]] if isinstance(imm, tuple) else imm, reloc) for op, a, b, imm, reloc in t.code]
        for t in self.tasks:
            base = len(insns)
            if not t.code or t.code[-1][0] not in (A.OP["DONE"], A.OP["JMP"], A.OP["PANIC"]):
                t.done()
            progs.append(A.Prog(t.node, t.flags, base))
            for op, a, b, imm, reloc in t.code:
                insns.append(A.Insn(op, a, (b + base) if reloc else b, imm))
        if len(insns) > 0xFFFF:
            raise ValueError("program too long")
        built = BuiltWorkload(self.nodes, progs, self.socks, insns, self.services, rows, dyn_max if rows is not None else 0)
        built.payloads = list(self.payloads)
        built.rpc_messages = list(self.rpc_messages)
        # what the rows were evaluated from (tests/golden/make_golden_async.py re-evaluates `contains` literally from these)
        built.panic_patterns = dict(self.panic_patterns)
        built.panic_text_of = {c: m for m, c in (codes.items() if rows is not None else ())}
        return built


def pingpong(n_nodes=4, rounds=64):
    """SURVEY.md §8d synthetic workload: pairs (1,2),(3,4).. of Endpoint ping-pong.

    Odd node = pinger: bind, sleep(1 s), R x {send_to(peer, 1, "ping"); recv_from(1); assert "pong"}.
    Even node = ponger: bind, R x {recv_from(1); assert "ping"; send_to(from, 1, "pong")}.
    Main (node 0) spawns one task per node in order and awaits every JoinHandle in order.
    The C twin is madsim_workload_pingpong() in the library; tests check both emit the same table.
    """
    if n_nodes % 2 or n_nodes < 2:
        raise ValueError("n_nodes must be even")
    wl = WorkloadBuilder()
    nodes = [wl.create_node() for _ in range(n_nodes)]
    addrs = [wl.addr(n, 1) for n in nodes]
    tasks = []
    for i, n in enumerate(nodes):
        t = wl.task(n)
        t.bind(addrs[i])
        if i % 2 == 0:
            t.sleep(secs=1)
            t.set(0, rounds)
            top = t.label()
            t.send_to(addrs[i], addrs[i + 1], 1, PING)
            t.recv_from(addrs[i], 1)
            t.assert_val(PONG)
            t.djnz(0, top)
        else:
            t.set(0, rounds)
            top = t.label()
            t.recv_from(addrs[i], 1)
            t.assert_val(PING)
            t.reply(addrs[i], 1, PONG)
            t.djnz(0, top)
        t.done()
        tasks.append(t)
    m = wl.main()
    for t in tasks:
        m.spawn(t)
    for t in tasks:
        m.join(t)
    m.done()
    return wl.build()


def raft_election(n_nodes=5, heartbeats=20, partitions=4):
    """BASELINE configs[2] shape: an n-node leader-election loop under NetSim partition injection.

    MadRaft itself is an external repository (only linked from README.md:78), so this is a stylised
    election with the same *event shape*: per node a follower/candidate/leader task driven by
    `timeout(election_timeout, ep.recv_from(..))` (staggered timeouts + a random candidate back-off), a voter task
    answering RequestVote on the same endpoint, leader heartbeats every 50 ms, and a supervisor that clogs / unclogs
    one node at a time at random moments (`NetSim::clog_node`, `sleep(gen_range(..))`).
    All traffic to a node's main task uses one tag (heartbeats carry 100+leader, vote grants carry 1), so every
    message is eventually consumed, as in a real single-mailbox Raft loop.  The run fails (panic verdict) if no leader
    was ever elected.
    """
    MAIN, VREQ = 1, 2
    GRANT = 1
    wl = WorkloadBuilder()
    nodes = [wl.create_node() for _ in range(n_nodes)]
    addrs = [wl.addr(n, 1) for n in nodes]
    need = n_nodes // 2                                   # grants needed besides the candidate's own vote
    mains, voters = [], []
    for i, n in enumerate(nodes):
        peers = [j for j in range(n_nodes) if j != i]
        t = wl.task(n)
        v = wl.task(n)                                     # voter: shares the endpoint (Endpoint is Clone), starts after bind
        t.bind(addrs[i])
        t.spawn(v)
        follower = t.label()
        t.recv_from_timeout(addrs[i], MAIN, ms=150 + 40 * i)
        cand_jump = len(t.code); t.jeq(A.VAL_TIMEOUT, 0)       # -> candidate (patched below)
        t.jmp(follower)                                    # heartbeat or stale grant: stay follower, timer restarts
        candidate = t.label()
        t.code[cand_jump][2] = candidate
        t.sleep_rand(lo_ms=0, ms=50)
        for j in peers:
            t.send_to(addrs[i], addrs[j], VREQ, i)
        t.set(1, need)
        collect = t.label()
        t.recv_from_timeout(addrs[i], MAIN, ms=100)
        t.jeq(A.VAL_TIMEOUT, follower)                     # split vote / partitioned: back to follower
        got_grant = len(t.code); t.jeq(GRANT, 0)           # patched: count it
        t.jmp(follower)                                    # someone else's heartbeat: step down
        t.code[got_grant][2] = t.label()
        t.djnz(1, collect)
        t.flag_add(0, 1)                                   # elected
        t.trace(0x200 + i)
        t.set(0, heartbeats)
        hb = t.label()
        for j in peers:
            t.send_to(addrs[i], addrs[j], MAIN, 100 + i)
        t.recv_from_timeout(addrs[i], MAIN, ms=50)         # heartbeat interval; drains late grants / rival heartbeats
        stay = len(t.code); t.jeq(A.VAL_TIMEOUT, 0)
        keep = len(t.code); t.jeq(GRANT, 0)
        t.jmp(follower)                                    # a rival leader's heartbeat: step down
        t.code[stay][2] = t.label(); t.code[keep][2] = t.label()
        t.djnz(0, hb)
        t.jmp(follower)                                    # term over: step down
        mains.append(t)
        top = v.label()
        v.recv_from(addrs[i], VREQ); v.reply(addrs[i], MAIN, GRANT); v.jmp(top)
        voters.append(v)
    m = wl.main()
    for t in mains:
        m.spawn(t)
    for k in range(partitions):
        victim = nodes[k % n_nodes]
        m.sleep_rand(lo_ms=0, secs=1)
        m.clog_node(victim, "both")
        m.sleep(ms=300)
        m.unclog_node(victim, "both")
    m.sleep(secs=2)
    m.panic_if_flag_lt(0, 1)
    m.done()
    return wl.build()


def kv_rpc(n_clients=4, n_ops=8):
    """BASELINE configs[3] shape: one etcd-style KV op = one connection (madsim-etcd-client/src/kv.rs:37-53: connect1 ->
    send request -> recv response).  The server binds, spawns the service's 1 s housekeeping tick task
    (service.rs:27-33), accepts in a loop and spawns one handler task per connection (server.rs:34-40); the handler
    receives the request and runs `service.timeout()` (service.rs:164-175): one `gen_bool(timeout_rate)` draw —
    config.loss_table[1], 0 by default — and on true a `sleep(gen_range(5 s..15 s))` before an error response."""
    REQ, RSP, ERR = 0x11, 0x22, 0x33
    wl = WorkloadBuilder()
    ns = wl.create_node()
    asv = wl.addr(ns, 2379)
    handler = wl.task(ns)
    handler.chan_recv(); handler.assert_val(REQ); handler.flag_add(0, 1)
    handler.rand_bool(1)
    slow = handler.label() + 3
    handler.jeq(1, slow)
    handler.chan_send(RSP); handler.done()
    assert handler.label() == slow
    handler.sleep_rand(lo_ms=5000, secs=15); handler.chan_send(ERR)
    tick = wl.task(ns)
    top = tick.label()
    tick.flag_add(1, 1); tick.sleep(secs=1); tick.jmp(top)
    srv = wl.task(ns)
    srv.bind(asv); srv.spawn(tick)
    top = srv.label()
    srv.accept1(asv); srv.spawn(handler, move_conn=True); srv.jmp(top)
    clients = []
    for i in range(n_clients):
        nc = wl.create_node()
        acl = wl.addr(nc, 1)
        c = wl.task(nc)
        c.bind(acl); c.sleep(ms=10); c.set(0, n_ops)
        top = c.label()
        c.connect1(acl, asv); c.assert_val(0); c.chan_send(REQ); c.chan_recv()
        ok = c.label() + 2
        c.jeq(RSP, ok); c.assert_val(ERR)
        c.chan_close(); c.djnz(0, top)
        clients.append(c)
    m = wl.main()
    m.spawn(srv)
    for c in clients:
        m.spawn(c)
    for c in clients:
        m.join(c)
    m.assert_flag(0, n_clients * n_ops)
    return wl.build()


def streaming_topology(n_compute=12, n_brokers=3, rounds=4):
    """BASELINE configs[4] shape: a 16-node streaming deployment (1 meta + 3 brokers + 12 compute nodes) built from the
    event shapes of the simulators such a system runs on — meta = etcd-style KV service over connect1/accept1 with its
    1 s tick task (madsim-etcd-client/src/server.rs:34-40, service.rs:27-33); brokers = typed-RPC handlers, one task per
    request (net/rpc.rs:152-179, the madsim-rdkafka sim_broker shape); compute nodes register with meta, then loop
    { produce to a broker with call_timeout; exchange a barrier datagram with the next compute node under timeout() }.
    A supervisor clogs one broker for a while and restarts another (init task re-binds).  Dozens of concurrent timers
    (timeouts' duplicate timers, ticks, backoffs) push the event heap into the HBM spill region."""
    REQ, RSP, ACK = 0x11, 0x22, 0x5A
    wl = WorkloadBuilder()
    meta = wl.create_node()
    a_meta = wl.addr(meta, 2379)
    h = wl.task(meta)
    h.chan_recv(); h.assert_val(REQ); h.flag_add(0, 1); h.chan_send(RSP)
    tick = wl.task(meta)
    top = tick.label()
    tick.flag_add(1, 1); tick.sleep(secs=1); tick.jmp(top)
    ms = wl.task(meta)
    ms.bind(a_meta); ms.spawn(tick)
    top = ms.label()
    ms.accept1(a_meta); ms.spawn(h, move_conn=True); ms.jmp(top)
    brokers = []
    for b in range(n_brokers):
        n = wl.create_node(); a = wl.addr(n, 9092)
        bh = wl.task(n)
        bh.sleep(ms=2 + b); bh.flag_add(2, 1); bh.rpc_reply(a, ACK)
        bs = wl.task(n, init=True)
        bs.bind(a)
        top = bs.label()
        bs.rpc_recv(a, 0); bs.spawn(bh, move_request=True); bs.jmp(top)
        brokers.append((n, a))
    nodes = [wl.create_node() for _ in range(n_compute)]
    addrs = [wl.addr(n, 5688) for n in nodes]
    computes = []
    for i in range(n_compute):
        c = wl.task(nodes[i])
        c.bind(addrs[i]); c.sleep(ms=5 + 7 * i)
        c.connect1(addrs[i], a_meta); c.assert_val(0); c.chan_send(REQ); c.chan_recv(); c.assert_val(RSP); c.chan_close()
        c.set(0, rounds)
        top = c.label()
        c.rpc_call(addrs[i], brokers[i % n_brokers][1], 0, i, timeout_ms=200)
        c.trace(700 + i)                                             # ACK or TIMEOUT: both are fine, both are observed
        c.send_to(addrs[i], addrs[(i + 1) % n_compute], 1, 0xB0 + i)
        c.recv_from_timeout(addrs[i], 1, ms=40)
        c.trace(800 + i)
        c.sleep_rand(lo_ms=0, ms=30)
        c.djnz(0, top)
        computes.append(c)
    m = wl.main()
    m.spawn(ms)
    for n, _ in brokers:
        m.build_node(n)
    for c in computes:
        m.spawn(c)
    m.sleep(ms=60); m.clog_node(brokers[0][0], "both")
    m.sleep(ms=150); m.unclog_node(brokers[0][0], "both")
    m.kill(brokers[1][0]); m.sleep(ms=50); m.restart(brokers[1][0])
    for c in computes:
        m.join(c)
    m.assert_flag(0, n_compute)
    return wl.build()


def streaming_topology_limits():
    """Capacities for streaming_topology: a small LDS heap quota with the bulk in the HBM spill region."""
    lim = A.Limits()
    lim.max_tasks = 28
    # (round 4: the every-class global-state build runs two waves per SIMD, and the LDS of the third workgroup holds 15 heap entries
    # per seed instead of 8: 4.65 against 4.13 G steps/s)
    # (round 6: 8-byte heap entries — MADSIM_STATE_NARROW_HEAP, every sleep / timeout of the workload is far below the 2.1 s horizon — so the
    # same LDS holds 31 entries, heap levels 0-4 complete: 5.31 against 4.72 G steps/s, profiles/r6_experiments.md)
    lim.heap_lds_slots, lim.heap_spill_slots = 31, 161
    lim.state_mem = A.STATE_NARROW_HEAP
    lim.mbox_regs, lim.mbox_msgs = 6, 5
    lim.max_conns, lim.chan_queue = 4, 1
    return lim


def raft_election_limits():
    """Device capacities the election loop needs (high-water marks over 40 000 seeds on the CPU oracle: timer heap 99
    — the duplicate timers of timeout() — so most of it lives in the HBM spill region; 67 dead recv registrations
    per socket, 8 queued messages).  The mailboxes live in global memory (Variant::G), so their
    capacity costs no LDS; rarer seeds still come back MADSIM_OVERFLOW and are re-run by run_batch_auto."""
    lim = A.Limits()
    lim.heap_lds_slots, lim.heap_spill_slots = 22, 234
    lim.mbox_regs, lim.mbox_msgs = 80, 10
    # a third of this workload's Timer::add calls re-register a pending Sleep (time/sleep.rs:51-53): kept as counts beside the
    # first entry (include/madsim_hip.h MADSIM_STATE_DEDUP_TIMERS; the layout itself stays on auto) — 25 % fewer global accesses
    # per step, the timer-fire phase a third shorter, +2-4 % measured (profiles/r3_experiments.md); 5e-4 of the seeds meet a tie
    # between different events and are run again by the kernel with the literal heap
    # Round 4: 32 seed lanes per wave on the global-state build.  This workload's rate is set by the memory system per lane access
    # (with every other lane idle it keeps 0.9 of its rate: tools/experiment/k_experiment.h EXP_HALF_LANES), so half the seeds per
    # wave at twice the LDS per seed — 22 of the ~20 distinct heap entries out of the spill region — wins: 9.25 G steps/s against
    # 8.44 with 64 lanes and 10 entries, 8.87 with 64 lanes, 16 entries and two waves per SIMD (profiles/r4_experiments.md).
    # Round 6: 8-byte heap entries (MADSIM_STATE_NARROW_HEAP; the workload's longest sleep is 2 s, inside the horizon) keep on FULL waves the
    # 20 entries in LDS that round 4 bought with 32 lanes and 22 wide ones — and the lanes are back: 9.83 against 9.2 G steps/s on one box
    # (32 lanes with 44 narrow entries: 9.72), profiles/r6_experiments.md.
    lim.lanes_per_wave = 0
    lim.state_mem = A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS | A.STATE_NARROW_HEAP
    return lim


def kv_rpc_limits():
    """High-water marks of the KV workload: 11 live tasks, 6 timers, 4 connections with 1 queued payload, datagram
    mailboxes unused (everything rides the reliable channel)."""
    lim = A.Limits()
    lim.max_tasks = 12
    lim.heap_lds_slots, lim.heap_spill_slots = 8, 0
    lim.mbox_regs, lim.mbox_msgs = A.LIMIT_NONE, A.LIMIT_NONE
    lim.max_conns, lim.chan_queue = 4, 1
    return lim


def timer_storm(n_tasks=24, rounds=16):
    """A timer-heavy workload for the HBM heap-spill path (BASELINE configs[4]: "event-heap HBM spill path"):
    n_tasks tasks each loop `sleep(gen_range(0..2 s))`, so the timer heap holds ~n_tasks entries at all times; with a
    small LDS quota most of every sift walks the [slot][lane] spill region in HBM."""
    wl = WorkloadBuilder()
    n = wl.create_node()
    tasks = []
    for _ in range(n_tasks):
        t = wl.task(n)
        t.set(0, rounds)
        top = t.label()
        t.sleep_rand(lo_ms=0, secs=2)
        t.djnz(0, top)
        tasks.append(t)
    m = wl.main()
    for t in tasks:
        m.spawn(t)
    for t in tasks:
        m.join(t)
    return wl.build()


def timer_storm_limits(lds_slots=4):
    lim = A.Limits()
    lim.heap_lds_slots, lim.heap_spill_slots = lds_slots, 64
    return lim


# ---- the workloads bench.py times: ONE definition, shared with the -m gpu parity tests ---------------------------------
BENCH_SEEDS_PER_GPU = 65536
BENCH_NODES, BENCH_ROUNDS = 4, 64
BENCH_WORKLOADS = ("pingpong", "raft", "kv", "timers", "topo")


def pingpong_bench_limits(heap_lds=4):
    """Tight capacities of the 4-node ping-pong (high-water marks on the oracle: 4 timers, 1 pending recv per socket,
    never a queued message).  Exceeding one shows up as verdict MADSIM_OVERFLOW, never as a different answer."""
    lim = A.Limits()
    lim.heap_lds_slots, lim.heap_spill_slots = heap_lds, 4 - heap_lds
    lim.mbox_regs, lim.mbox_msgs = 1, A.LIMIT_NONE
    return lim


def bench_case(name="pingpong", nodes=BENCH_NODES, rounds=BENCH_ROUNDS, heap_lds=4):
    """(workload, limits, description) exactly as bench.py runs `--workload name`; tests/test_gpu_parity.py oracle-checks
    these same objects, so the configuration behind the headline number is the configuration that is verified."""
    if name == "pingpong":
        return (pingpong(nodes, rounds), pingpong_bench_limits(heap_lds),
                f"{nodes}-node ping-pong, R={rounds}, Config::default()")
    if name == "raft":
        return raft_election(), raft_election_limits(), "5-node election loop with partition injection (configs[2] shape)"
    if name == "timers":
        return (timer_storm(), timer_storm_limits(heap_lds),
                f"timer storm: 24 tasks x sleep(gen_range(0..2 s)), heap_lds={heap_lds} (HBM heap-spill path)")
    if name == "topo":
        return (streaming_topology(), streaming_topology_limits(),
                "16-node streaming topology: KV meta + typed-RPC brokers + 12 compute nodes (configs[4] shape)")
    if name == "kv":
        return kv_rpc(), kv_rpc_limits(), "etcd-style KV ops over connect1/accept1 (configs[3] shape)"
    raise ValueError(f"unknown bench workload {name!r}")
