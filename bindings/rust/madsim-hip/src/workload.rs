//! The workload DSL: one method per reference API call, one 8-byte instruction per call (`enum madsim_op`,
//! `include/madsim_hip.h`).  Mirrors `madsim::WorkloadBuilder` / `madsim::Task` of `include/madsim_hip.hpp`.
use madsim_hip_sys as sys;
use std::time::Duration;

/// Index of a task program (program 0 = the future handed to `block_on`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct TaskId(pub u8);
/// Index into the socket-address table.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Addr(pub u8);

#[derive(Clone, Copy)]
struct Ins {
    insn: sys::madsim_insn_t,
    reloc: bool,
}

/// One async block (`node.spawn(async move { ... })`).
pub struct TaskBuilder {
    index: u8,
    node: u8,
    flags: u8,
    code: Vec<Ins>,
}

fn dur_parts(d: Duration) -> (u16, u32) {
    assert!(d.as_secs() <= u16::MAX as u64, "durations are encoded as u16 seconds + u32 nanoseconds");
    (d.as_secs() as u16, d.subsec_nanos())
}

impl TaskBuilder {
    fn emit(&mut self, op: u8, a: u8, b: u16, imm: u32, reloc: bool) -> &mut Self {
        self.code.push(Ins { insn: sys::madsim_insn_t { op, a, b, imm }, reloc });
        self
    }
    pub fn id(&self) -> TaskId {
        TaskId(self.index)
    }
    /// Position of the next instruction (a jump target inside this task).
    pub fn label(&self) -> u16 {
        self.code.len() as u16
    }
    // ---- task / control ------------------------------------------------------------------------------------------------
    pub fn done(&mut self) -> &mut Self { self.emit(sys::MS_OP_DONE, 0, 0, 0, false) }
    /// `NodeHandle::spawn` / `task::spawn` (task/mod.rs:607-654)
    pub fn spawn(&mut self, t: TaskId) -> &mut Self { self.emit(sys::MS_OP_SPAWN, t.0, 0, 0, false) }
    /// `handle.await.unwrap()` (task/join.rs:59-72); `expect_err`: `.unwrap_err()`
    pub fn join(&mut self, t: TaskId, expect_err: bool) -> &mut Self { self.emit(sys::MS_OP_JOIN, t.0, expect_err as u16, 0, false) }
    pub fn abort(&mut self, t: TaskId) -> &mut Self { self.emit(sys::MS_OP_ABORT, t.0, 0, 0, false) }
    pub fn yield_now(&mut self) -> &mut Self { self.emit(sys::MS_OP_YIELD, 0, 0, 0, false) }
    /// `panic!()` with message code `code` (what `restart_on_panic_matching` names)
    pub fn panic(&mut self, code: u8) -> &mut Self { self.emit(sys::MS_OP_PANIC, 0, 0, code as u32, false) }
    pub fn set(&mut self, reg: u8, v: u16) -> &mut Self { self.emit(sys::MS_OP_SET, reg, 0, v as u32, false) }
    pub fn djnz(&mut self, reg: u8, target: u16) -> &mut Self { self.emit(sys::MS_OP_DJNZ, reg, target, 0, true) }
    pub fn jmp(&mut self, target: u16) -> &mut Self { self.emit(sys::MS_OP_JMP, 0, target, 0, true) }
    pub fn jeq(&mut self, value: u32, target: u16) -> &mut Self { self.emit(sys::MS_OP_JEQ, 0, target, value, true) }
    /// an observable side effect (folded into `obs_hash`)
    pub fn trace(&mut self, v: u32) -> &mut Self { self.emit(sys::MS_OP_TRACE, 0, 0, v, false) }
    /// observe `Instant` elapsed since the main future's first poll
    pub fn trace_instant(&mut self) -> &mut Self { self.emit(sys::MS_OP_TRACE_TIME, 1, 0, 0, false) }
    /// observe the last received / drawn value
    pub fn trace_val(&mut self) -> &mut Self { self.emit(sys::MS_OP_TRACE_TIME, 2, 0, 0, false) }
    /// `val = madsim::rand::random::<u32>()` (one `GlobalRng::with`)
    pub fn random_u32(&mut self) -> &mut Self { self.emit(sys::MS_OP_RANDOM, 0, 0, 0, false) }
    /// `for _ in 0..n { body }` on loop register `reg`
    pub fn repeat(&mut self, reg: u8, n: u16, body: impl FnOnce(&mut Self)) -> &mut Self {
        self.set(reg, n);
        let top = self.label();
        body(self);
        self.djnz(reg, top)
    }
    // ---- time ------------------------------------------------------------------------------------------------------------
    pub fn sleep(&mut self, d: Duration) -> &mut Self { let (s, ns) = dur_parts(d); self.emit(sys::MS_OP_SLEEP, 0, s, ns, false) }
    pub fn mark(&mut self) -> &mut Self { self.emit(sys::MS_OP_MARK, 0, 0, 0, false) }
    pub fn sleep_until(&mut self, after_mark: Duration) -> &mut Self { let (s, ns) = dur_parts(after_mark); self.emit(sys::MS_OP_SLEEP_UNTIL, 0, s, ns, false) }
    pub fn assert_elapsed_eq(&mut self, d: Duration) -> &mut Self { let (s, ns) = dur_parts(d); self.emit(sys::MS_OP_ASSERT_ELAPSED, 0, s, ns, false) }
    pub fn advance(&mut self, d: Duration) -> &mut Self { let (s, ns) = dur_parts(d); self.emit(sys::MS_OP_ADVANCE, 0, s, ns, false) }
    // ---- datagram Endpoint API (net/endpoint.rs) ------------------------------------------------------------------------------
    /// `Endpoint::bind(addr).await.unwrap()`
    pub fn bind(&mut self, a: Addr) -> &mut Self { self.emit(sys::MS_OP_BIND, a.0, 0, 0, false) }
    /// `ep.send_to(dst, tag, payload).await.unwrap()`
    pub fn send_to(&mut self, ep: Addr, dst: Addr, tag: u8, payload: u32) -> &mut Self { self.emit(sys::MS_OP_SEND, ep.0, ((tag as u16) << 8) | dst.0 as u16, payload, false) }
    /// `ep.send_to(from, tag, payload).await.unwrap()` — answer the last received datagram
    pub fn reply(&mut self, ep: Addr, tag: u8, payload: u32) -> &mut Self { self.emit(sys::MS_OP_REPLY, ep.0, (tag as u16) << 8, payload, false) }
    /// `(val, from) = ep.recv_from(tag).await.unwrap()`
    pub fn recv_from(&mut self, ep: Addr, tag: u8) -> &mut Self { self.emit(sys::MS_OP_RECV, ep.0, (tag as u16) << 8, 0, false) }
    pub fn assert_val(&mut self, v: u32) -> &mut Self { self.emit(sys::MS_OP_ASSERT_VAL, 0, 0, v, false) }
    pub fn close(&mut self, ep: Addr) -> &mut Self { self.emit(sys::MS_OP_CLOSE, ep.0, 0, 0, false) }
    /// `timeout(d, ep.recv_from(tag)).await`: `val = MADSIM_VAL_TIMEOUT` on `Err(Elapsed)`
    pub fn recv_from_timeout(&mut self, ep: Addr, tag: u8, d: Duration) -> &mut Self {
        assert!(d.as_secs() <= 255);
        self.emit(sys::MS_OP_RECV_TIMEOUT, ep.0, ((tag as u16) << 8) | d.as_secs() as u16, d.subsec_nanos(), false)
    }
    // ---- supervisor ----------------------------------------------------------------------------------------------------------
    pub fn kill(&mut self, node: u8) -> &mut Self { self.emit(sys::MS_OP_KILL, node, 0, 0, false) }
    pub fn restart(&mut self, node: u8) -> &mut Self { self.emit(sys::MS_OP_RESTART, node, 0, 0, false) }
    pub fn pause(&mut self, node: u8) -> &mut Self { self.emit(sys::MS_OP_PAUSE, node, 0, 0, false) }
    pub fn resume(&mut self, node: u8) -> &mut Self { self.emit(sys::MS_OP_RESUME, node, 0, 0, false) }
    pub fn clog_node(&mut self, node: u8) -> &mut Self { self.emit(sys::MS_OP_CLOG_NODE, node, 3, 0, false) }
    pub fn unclog_node(&mut self, node: u8) -> &mut Self { self.emit(sys::MS_OP_UNCLOG_NODE, node, 3, 0, false) }
    pub fn clog_link(&mut self, src: u8, dst: u8) -> &mut Self { self.emit(sys::MS_OP_CLOG_LINK, src, dst as u16, 0, false) }
    pub fn unclog_link(&mut self, src: u8, dst: u8) -> &mut Self { self.emit(sys::MS_OP_UNCLOG_LINK, src, dst as u16, 0, false) }
    /// `NetSim::current().update_config(|c| c.send_latency = latency_table[index])` (`net/mod.rs:138-141`)
    pub fn set_latency(&mut self, index: u8) -> &mut Self { self.emit(sys::MS_OP_SET_LATENCY, index, 0, 0, false) }
    // ---- shared flags (the Arc<AtomicUsize> the reference's tests observe) ------------------------------------------------------
    pub fn flag_store(&mut self, flag: u8, v: u32) -> &mut Self { self.emit(sys::MS_OP_GSET, flag, 0, v, false) }
    pub fn flag_add(&mut self, flag: u8, v: u32) -> &mut Self { self.emit(sys::MS_OP_GADD, flag, 0, v, false) }
    pub fn assert_flag(&mut self, flag: u8, v: u32) -> &mut Self { self.emit(sys::MS_OP_ASSERT_G, flag, 0, v, false) }
    // ---- reliable channel, typed RPC --------------------------------------------------------------------------------------------
    pub fn connect1(&mut self, ep: Addr, dst: Addr) -> &mut Self { self.emit(sys::MS_OP_CONNECT, ep.0, dst.0 as u16, 0, false) }
    pub fn accept1(&mut self, ep: Addr) -> &mut Self { self.emit(sys::MS_OP_ACCEPT, ep.0, 0, 0, false) }
    pub fn chan_send(&mut self, payload: u32) -> &mut Self { self.emit(sys::MS_OP_CSEND, 0, 0, payload, false) }
    pub fn chan_recv(&mut self) -> &mut Self { self.emit(sys::MS_OP_CRECV, 0, 0, 0, false) }
    pub fn chan_close(&mut self) -> &mut Self { self.emit(sys::MS_OP_CCLOSE, 0, 0, 0, false) }
    /// `ep.call(dst, req).await` / `ep.call_timeout(dst, req, d).await` (net/rpc.rs:96-131); `req_id` = `R::ID - 0x80`
    pub fn rpc_call(&mut self, ep: Addr, dst: Addr, req_id: u8, code: u8, timeout: Option<Duration>) -> &mut Self {
        let ms = timeout.map_or(0, |d| d.as_millis() as u32);
        assert!(ms < (1 << 24));
        self.emit(sys::MS_OP_RPC_CALL, ep.0, (((sys::MADSIM_TAG_RPC_FIRST as u16) + req_id as u16) << 8) | dst.0 as u16, (ms << 8) | code as u32, false)
    }
    pub fn rpc_recv(&mut self, ep: Addr, req_id: u8) -> &mut Self { self.emit(sys::MS_OP_RECV, ep.0, ((sys::MADSIM_TAG_RPC_FIRST as u16) + req_id as u16) << 8, 0, false) }
    pub fn rpc_reply(&mut self, ep: Addr, code: u8) -> &mut Self { self.emit(sys::MS_OP_RPC_REPLY, ep.0, 0, code as u32, false) }
    // ---- NetSim::global_ipvs() at run time (net/ipvs.rs:50-85); `service` = the index `ipvs_service` returned -------------------
    pub fn ipvs_add_service(&mut self, service: u8) -> &mut Self { self.emit(sys::MS_OP_IPVS, sys::MADSIM_IPVS_ADD_SERVICE as u8, service as u16, 0, false) }
    pub fn ipvs_del_service(&mut self, service: u8) -> &mut Self { self.emit(sys::MS_OP_IPVS, sys::MADSIM_IPVS_DEL_SERVICE as u8, service as u16, 0, false) }
    pub fn ipvs_add_server(&mut self, service: u8, server: Addr) -> &mut Self { self.emit(sys::MS_OP_IPVS, sys::MADSIM_IPVS_ADD_SERVER as u8, service as u16, server.0 as u32, false) }
    pub fn ipvs_del_server(&mut self, service: u8, server: Addr) -> &mut Self { self.emit(sys::MS_OP_IPVS, sys::MADSIM_IPVS_DEL_SERVER as u8, service as u16, server.0 as u32, false) }
}

/// Owns the tables a `madsim_workload_t` points into.
pub struct Workload {
    pub nodes: Vec<sys::madsim_node_t>,
    pub progs: Vec<sys::madsim_prog_t>,
    pub socks: Vec<sys::madsim_sock_t>,
    pub insns: Vec<sys::madsim_insn_t>,
    /// IPVS virtual services (net/ipvs.rs): address entry + real servers in add_server order
    pub services: Vec<sys::madsim_service_t>,
}

impl Workload {
    /// The C view; valid while `self` is alive and unmodified.
    pub fn raw(&self) -> sys::madsim_workload_t {
        sys::madsim_workload_t {
            n_nodes: (self.nodes.len() - 1) as u32,
            n_progs: self.progs.len() as u32,
            n_socks: self.socks.len() as u32,
            n_insns: self.insns.len() as u32,
            nodes: self.nodes.as_ptr(),
            progs: self.progs.as_ptr(),
            socks: self.socks.as_ptr(),
            insns: self.insns.as_ptr(),
            n_services: self.services.len() as u32,
            panic_dyn_max: 0,
            services: if self.services.is_empty() { std::ptr::null() } else { self.services.as_ptr() },
            panic_match: std::ptr::null(),
        }
    }
}

pub struct WorkloadBuilder {
    nodes: Vec<sys::madsim_node_t>,
    socks: Vec<sys::madsim_sock_t>,
    tasks: Vec<TaskBuilder>,
    payloads: Vec<Vec<u8>>,
    services: Vec<sys::madsim_service_t>,
}

impl Default for WorkloadBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl WorkloadBuilder {
    pub fn new() -> Self {
        WorkloadBuilder {
            nodes: vec![sys::madsim_node_t { flags: 0, n_match: 0, r#match: [0; 2] }],   // node 0 = "madsim-main"
            socks: Vec::new(),
            tasks: vec![TaskBuilder { index: 0, node: 0, flags: 0, code: Vec::new() }],
            payloads: Vec::new(),
            services: Vec::new(),
        }
    }
    /// The future handed to `block_on` (program 0, on node 0).
    pub fn main(&mut self) -> &mut TaskBuilder {
        &mut self.tasks[0]
    }
    pub fn task_mut(&mut self, t: TaskId) -> &mut TaskBuilder {
        &mut self.tasks[t.0 as usize]
    }
    /// `Handle::create_node().ip(10.0.0.<id>)[.restart_on_panic()].build()`; returns the node id (1..)
    pub fn create_node(&mut self, restart_on_panic: bool) -> u8 {
        assert!(self.nodes.len() <= 31, "at most 31 nodes");
        let flags = if restart_on_panic { sys::MADSIM_NODE_RESTART_ON_PANIC as u8 } else { 0 };
        self.nodes.push(sys::madsim_node_t { flags, n_match: 0, r#match: [0; 2] });
        (self.nodes.len() - 1) as u8
    }
    /// `10.0.0.<node>:port`
    pub fn addr(&mut self, node: u8, port: u16) -> Addr {
        self.socks.push(sys::madsim_sock_t { node, kind: sys::MADSIM_ADDR_IP as u8, port });
        Addr((self.socks.len() - 1) as u8)
    }
    /// A virtual service address that belongs to no node ("1.1.1.<ip_id>:<port>"): a destination only.
    pub fn virtual_addr(&mut self, ip_id: u8, port: u16) -> Addr {
        assert!(ip_id >= 1 && port >= 1);
        self.socks.push(sys::madsim_sock_t { node: ip_id, kind: sys::MADSIM_ADDR_VIRTUAL as u8, port });
        Addr((self.socks.len() - 1) as u8)
    }
    /// `ipvs.add_service(ServiceAddr::Tcp(vaddr), RoundRobin)` + one `add_server` per entry (net/ipvs.rs:50-85), before any task runs
    /// (`absent`: only the address is declared — the service exists once a task calls `ipvs_add_service`).  Returns the service index.
    pub fn ipvs_service(&mut self, vaddr: Addr, servers: &[Addr], absent: bool) -> u8 {
        assert!(self.services.len() < sys::MADSIM_MAX_SERVICES as usize && servers.len() <= 6 && !(absent && !servers.is_empty()));
        let n = if absent { sys::MADSIM_SERVICE_ABSENT as u8 } else { servers.len() as u8 };
        let mut s = sys::madsim_service_t { vaddr: vaddr.0, n_servers: n, servers: [0; 6] };
        for (i, a) in servers.iter().enumerate() {
            s.servers[i] = a.0;
        }
        self.services.push(s);
        (self.services.len() - 1) as u8
    }
    /// A new task program on `node` (`node.spawn(async move { .. })` once a `spawn` instruction names it).
    pub fn task(&mut self, node: u8) -> TaskId {
        assert!(self.tasks.len() < 255, "at most 255 task programs");
        let index = self.tasks.len() as u8;
        self.tasks.push(TaskBuilder { index, node, flags: 0, code: Vec::new() });
        TaskId(index)
    }
    /// Byte strings never steer the simulation (a payload can only be compared), so they are interned: equal bytes <=> equal value.
    pub fn payload(&mut self, data: &[u8]) -> u32 {
        if let Some(i) = self.payloads.iter().position(|p| p == data) {
            return 0x4000_0000 + i as u32;
        }
        self.payloads.push(data.to_vec());
        0x4000_0000 + (self.payloads.len() - 1) as u32
    }
    /// The bytes behind an interned payload value (used by the real-madsim interpreter).
    pub fn payload_table(&self) -> Vec<Vec<u8>> {
        self.payloads.clone()
    }
    pub fn build(mut self) -> Workload {
        let mut w = Workload { nodes: self.nodes, progs: Vec::new(), socks: self.socks, insns: Vec::new(), services: self.services };
        for t in self.tasks.iter_mut() {
            let base = w.insns.len() as u16;
            let ends = t.code.last().map_or(false, |i| i.insn.op == sys::MS_OP_DONE || i.insn.op == sys::MS_OP_JMP || i.insn.op == sys::MS_OP_PANIC);
            if !ends {
                t.done();
            }
            w.progs.push(sys::madsim_prog_t { node: t.node, flags: t.flags, entry: base });
            for i in &t.code {
                let mut insn = i.insn;
                if i.reloc {
                    insn.b += base;
                }
                w.insns.push(insn);
            }
        }
        w
    }
}

pub const PING: u32 = 0x676E_6970; // b"ping" as a little-endian word, as madsim_workload_pingpong / workload.py::pingpong
pub const PONG: u32 = 0x676E_6F70;

/// SURVEY.md §8d workload: N nodes `10.0.0.i`, pairs (1,2),(3,4)..; pinger = bind, sleep(1 s), R x {send ping, recv pong};
/// ponger = bind, R x {recv ping, reply pong}; main spawns in node order and awaits the handles in order.  Instruction for
/// instruction the table `madsim_workload_pingpong` (C) and `madsim_amd.workload.pingpong` (Python) build.
pub fn pingpong(n_nodes: u8, rounds: u16) -> Workload {
    pingpong_with(n_nodes, rounds, false)
}

/// `pingpong` whose main future ends with the fingerprint tail of tools/ref_twin (`trace_instant(); random_u32(); trace_val()`):
/// the elapsed time, then one trailing draw whose value depends on every draw before it.
pub fn pingpong_twin(n_nodes: u8, rounds: u16) -> Workload {
    pingpong_with(n_nodes, rounds, true)
}

fn pingpong_with(n_nodes: u8, rounds: u16, tail: bool) -> Workload {
    assert!(n_nodes >= 2 && n_nodes % 2 == 0 && n_nodes <= 30 && rounds > 0);
    let mut wl = WorkloadBuilder::new();
    let nodes: Vec<u8> = (0..n_nodes).map(|_| wl.create_node(false)).collect();
    let addrs: Vec<Addr> = nodes.iter().map(|&n| wl.addr(n, 1)).collect();
    let tasks: Vec<TaskId> = nodes.iter().map(|&n| wl.task(n)).collect();
    for &t in &tasks {
        wl.main().spawn(t);
    }
    for &t in &tasks {
        wl.main().join(t, false);
    }
    if tail {
        wl.main().trace_instant().random_u32().trace_val();
    }
    for i in 0..n_nodes as usize {
        let (me, t) = (addrs[i], wl.task_mut(tasks[i]));
        t.bind(me);
        if i % 2 == 0 {
            let peer = addrs[i + 1];
            t.sleep(Duration::from_secs(1));
            t.repeat(0, rounds, |t| {
                t.send_to(me, peer, 1, PING).recv_from(me, 1).assert_val(PONG);
            });
        } else {
            t.repeat(0, rounds, |t| {
                t.recv_from(me, 1).assert_val(PING).reply(me, 1, PONG);
            });
        }
    }
    wl.build()
}
