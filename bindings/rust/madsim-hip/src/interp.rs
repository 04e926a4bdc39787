//! Runs a workload table on REAL madsim through its public API — the other half of "one Rust source, two executors":
//! `Builder::run_workload` hands the table to the GPU runner, `interp::block_on` interprets the same table instruction by
//! instruction with the reference's own `Endpoint`, `time::sleep`, `task::spawn`, `JoinHandle` ... inside a madsim `Runtime`.
//! tools/ref_twin uses it to run `workload::pingpong_twin` on the reference executor and print the fingerprint the oracle's
//! run of the very same table must reproduce.
//!
//! Covers the base ops (datagram Endpoint API, sleeps, spawn / join / abort / yield, loop and assert glue, shared flags,
//! clogs, kill / restart / pause / resume, observations).  Ops outside that set panic with "interp: unsupported op": they have
//! hand-written twins in tools/ref_twin/src/main.rs instead.  Only built with `--features madsim` and `--cfg madsim`.
use crate::workload::Workload;
use madsim::net::{Endpoint, NetSim};
use madsim::runtime::{Handle, NodeHandle};
use madsim::task::JoinHandle;
use madsim::time::{self, Duration, Instant};
use madsim_hip_sys as sys;
use std::collections::HashMap;
use std::future::Future;
use std::net::SocketAddr;
use std::pin::Pin;
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::{Arc, Mutex};

/// What a run made observable (the repo folds `obs` into `obs_hash`, compare.py diffs the list itself).
#[derive(Clone, Debug, Default)]
pub struct Observed {
    pub obs: Vec<u64>,
    pub elapsed_ns: u64,
    pub msg_count: u64,
}

struct Shared {
    nodes: Vec<sys::madsim_node_t>,
    progs: Vec<sys::madsim_prog_t>,
    socks: Vec<sys::madsim_sock_t>,
    insns: Vec<sys::madsim_insn_t>,
    payloads: Vec<Vec<u8>>,
    lat_table: Vec<std::ops::Range<Duration>>,
    node_handles: Mutex<Vec<Option<NodeHandle>>>,
    join_handles: Mutex<Vec<Option<JoinHandle<()>>>>,
    flags: [AtomicUsize; 4],
    obs: Mutex<Vec<u64>>,
    t0: Mutex<Option<Instant>>,
}

impl Shared {
    fn addr(&self, a: u8) -> SocketAddr {
        let s = &self.socks[a as usize];
        let ip = match s.kind as u32 {
            sys::MADSIM_ADDR_UNSPECIFIED => "0.0.0.0".to_string(),
            sys::MADSIM_ADDR_LOOPBACK => "127.0.0.1".to_string(),
            _ => format!("10.0.0.{}", s.node),
        };
        format!("{ip}:{}", s.port).parse().unwrap()
    }
    /// the bytes on the wire of a payload value: an interned byte string, or the value's four little-endian bytes
    fn bytes_of(&self, v: u32) -> Vec<u8> {
        if v >= 0x4000_0000 && ((v - 0x4000_0000) as usize) < self.payloads.len() {
            return self.payloads[(v - 0x4000_0000) as usize].clone();
        }
        v.to_le_bytes().to_vec()
    }
    fn value_of(&self, b: &[u8]) -> u32 {
        if let Some(i) = self.payloads.iter().position(|p| p.as_slice() == b) {
            return 0x4000_0000 + i as u32;
        }
        let mut w = [0u8; 4];
        w[..b.len().min(4)].copy_from_slice(&b[..b.len().min(4)]);
        u32::from_le_bytes(w)
    }
    fn observe(&self, v: u64) {
        self.obs.lock().unwrap().push(v);
    }
}

fn dur(b: u16, imm: u32) -> Duration {
    Duration::new(b as u64, imm)
}

type Task = Pin<Box<dyn Future<Output = ()> + Send + 'static>>;

/// One instance of task program `prog` (`async move { .. }` handed to `spawn`).
fn run_task(sh: Arc<Shared>, prog: usize) -> Task {
    Box::pin(async move {
        let my_node = sh.progs[prog].node;
        let mut pc = sh.progs[prog].entry as usize;
        let mut cnt = [0u16; 2];
        let mut val: u32 = 0;
        let mut from: Option<SocketAddr> = None;
        let mut eps: HashMap<u8, Endpoint> = HashMap::new();
        let mut buf = vec![0u8; 4096];
        loop {
            let i = sh.insns[pc];
            let (a, b, imm) = (i.a, i.b, i.imm);
            pc += 1;
            match i.op {
                sys::MS_OP_DONE => return,
                sys::MS_OP_SPAWN => {
                    let target = a as usize;
                    let node = sh.progs[target].node;
                    let fut = run_task(sh.clone(), target);
                    // another node's program: NodeHandle::spawn; this node's: task::spawn (task/mod.rs:592-599)
                    let jh = if node == my_node {
                        madsim::task::spawn(fut)
                    } else {
                        let nh = sh.node_handles.lock().unwrap()[node as usize].clone().expect("node");
                        nh.spawn(fut)
                    };
                    sh.join_handles.lock().unwrap()[target] = Some(jh);
                }
                sys::MS_OP_JOIN => {
                    let jh = sh.join_handles.lock().unwrap()[a as usize].take().expect("join: no handle");
                    let r = jh.await;
                    if b & 1 == 1 { assert!(r.is_err()); } else { r.unwrap(); }
                }
                sys::MS_OP_ABORT => {
                    if let Some(jh) = sh.join_handles.lock().unwrap()[a as usize].as_ref() { jh.abort(); }
                }
                sys::MS_OP_YIELD => madsim::task::yield_now().await,
                sys::MS_OP_PANIC => panic!("{}", imm),
                sys::MS_OP_SET => cnt[(a & 1) as usize] = imm as u16,
                sys::MS_OP_DJNZ => {
                    let r = (a & 1) as usize;
                    cnt[r] = cnt[r].wrapping_sub(1);
                    if cnt[r] != 0 { pc = b as usize; }
                }
                sys::MS_OP_JMP => pc = b as usize,
                sys::MS_OP_JEQ => { if val == imm { pc = b as usize; } }
                sys::MS_OP_TRACE => sh.observe(imm as u64 + if b & 1 == 1 { cnt[(a & 1) as usize] as u64 } else { 0 }),
                sys::MS_OP_TRACE_TIME => match a {
                    1 => {
                        let t0: Option<Instant> = *sh.t0.lock().unwrap();
                        sh.observe(t0.expect("t0").elapsed().as_nanos() as u64)
                    }
                    2 => sh.observe(val as u64),
                    _ => panic!("interp: unsupported op TRACE_TIME a=0 (SystemTime needs the base-time draw)"),
                },
                sys::MS_OP_RANDOM if a == 0 => val = madsim::rand::random::<u32>(),
                sys::MS_OP_SLEEP => time::sleep(dur(b, imm)).await,
                sys::MS_OP_BIND => {
                    let ep = Endpoint::bind(sh.addr(a)).await.unwrap();
                    eps.insert(a, ep);
                }
                sys::MS_OP_SEND => {
                    let (tag, dst) = ((b >> 8) as u64, (b & 0xff) as u8);
                    eps[&a].send_to(sh.addr(dst), tag, &sh.bytes_of(imm)).await.unwrap();
                }
                sys::MS_OP_REPLY => {
                    eps[&a].send_to(from.expect("reply before recv"), (b >> 8) as u64, &sh.bytes_of(imm)).await.unwrap();
                }
                sys::MS_OP_RECV => {
                    let (len, f) = eps[&a].recv_from((b >> 8) as u64, &mut buf).await.unwrap();
                    val = sh.value_of(&buf[..len]);
                    from = Some(f);
                }
                sys::MS_OP_ASSERT_VAL => assert_eq!(val, imm),
                sys::MS_OP_CLOSE => { eps.remove(&a); }
                sys::MS_OP_GSET => sh.flags[(a & 3) as usize].store(imm as usize, Ordering::Relaxed),
                sys::MS_OP_GADD => { sh.flags[(a & 3) as usize].fetch_add(imm as usize, Ordering::Relaxed); }
                sys::MS_OP_ASSERT_G => assert_eq!(sh.flags[(a & 3) as usize].load(Ordering::Relaxed), imm as usize),
                sys::MS_OP_KILL => Handle::current().kill(sh.node_handles.lock().unwrap()[a as usize].as_ref().expect("node").id()),
                sys::MS_OP_RESTART => Handle::current().restart(sh.node_handles.lock().unwrap()[a as usize].as_ref().expect("node").id()),
                sys::MS_OP_PAUSE => Handle::current().pause(sh.node_handles.lock().unwrap()[a as usize].as_ref().expect("node").id()),
                sys::MS_OP_RESUME => Handle::current().resume(sh.node_handles.lock().unwrap()[a as usize].as_ref().expect("node").id()),
                sys::MS_OP_CLOG_NODE | sys::MS_OP_UNCLOG_NODE => {
                    let id = sh.node_handles.lock().unwrap()[a as usize].as_ref().expect("node").id();
                    let net = NetSim::current();
                    let clog = i.op == sys::MS_OP_CLOG_NODE;
                    if b & 1 == 1 { if clog { net.clog_node_in(id) } else { net.unclog_node_in(id) } }
                    if b & 2 == 2 { if clog { net.clog_node_out(id) } else { net.unclog_node_out(id) } }
                }
                sys::MS_OP_SET_LATENCY => {
                    let r = sh.lat_table[a as usize].clone();
                    NetSim::current().update_config(|c| c.send_latency = r);
                }
                other => panic!("interp: unsupported op {other}"),
            }
        }
    })
}

/// The future to hand to `Runtime::block_on`: creates the nodes (in table order, `10.0.0.<id>`), runs program 0 and returns
/// what the run made observable.  Panics of the workload propagate as panics of the future (= madsim's own behaviour).
pub async fn main_future(w: &Workload, payloads: Vec<Vec<u8>>) -> Observed {
    main_future_with(w, payloads, Vec::new()).await
}

/// `main_future` for a workload that calls `set_latency(i)`: `latency_table[i]` is what `NetConfig::latency_table` hands the GPU runner.
pub async fn main_future_with(w: &Workload, payloads: Vec<Vec<u8>>, latency_table: Vec<std::ops::Range<Duration>>) -> Observed {
    let sh = Arc::new(Shared {
        nodes: w.nodes.clone(),
        progs: w.progs.clone(),
        socks: w.socks.clone(),
        insns: w.insns.clone(),
        payloads,
        lat_table: latency_table,
        node_handles: Mutex::new((0..w.nodes.len()).map(|_| None).collect()),
        join_handles: Mutex::new((0..w.progs.len()).map(|_| None).collect()),
        flags: [AtomicUsize::new(0), AtomicUsize::new(0), AtomicUsize::new(0), AtomicUsize::new(0)],
        obs: Mutex::new(Vec::new()),
        t0: Mutex::new(None),
    });
    *sh.t0.lock().unwrap() = Some(Instant::now());
    let h = Handle::current();
    for id in 1..sh.nodes.len() {
        let mut nb = h.create_node();
        if sh.nodes[id].flags as u32 & sys::MADSIM_NODE_NO_IP == 0 {
            nb = nb.ip(format!("10.0.0.{id}").parse().unwrap());
        }
        if sh.nodes[id].flags as u32 & sys::MADSIM_NODE_RESTART_ON_PANIC != 0 {
            nb = nb.restart_on_panic();
        }
        sh.node_handles.lock().unwrap()[id] = Some(nb.build());
    }
    run_task(sh.clone(), 0).await;
    let t0: Option<Instant> = *sh.t0.lock().unwrap();
    let elapsed_ns = t0.unwrap().elapsed().as_nanos() as u64;
    let msg_count = NetSim::current().stat().msg_count;
    let obs = std::mem::take(&mut *sh.obs.lock().unwrap());
    Observed { obs, elapsed_ns, msg_count }
}
