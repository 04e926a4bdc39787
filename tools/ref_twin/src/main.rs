//! madsim-ref-twin — the reference side of the oracle pin.
//!
//! Runs REAL madsim on the workloads of SURVEY.md §8d (and the repo's golden / lifecycle workloads) and prints one JSON
//! line per seed.  Everything printed is observable through madsim's public API:
//!   * `elapsed_ns`  — `madsim::time::Instant` elapsed since the first poll of the main future (clock 0),
//!   * `msg_count`   — `NetSim::current().stat().msg_count`            (net/mod.rs:133-135, network.rs:99-105,265),
//!   * `obs`         — every value the workload made observable, in execution order (the repo's MS_OP_TRACE /
//!                     MS_OP_TRACE_TIME), ending with ONE trailing `madsim::rand::random::<u32>()` — a draw whose value
//!                     depends on every RNG call before it, so it pins the draw count and the generator state,
//!   * `verdict`     — pass / panic / deadlock / time-limit, from the panic message of `block_on` (task/mod.rs:250-258).
//! With `--features rng-log` (madsim patched with patches/expose_rng_log.patch) `log_hex` carries the raw determinism
//! log of rand.rs:64-88 for byte-level diffs against `madsim_hip_trace_seed` / the oracle.
//!
//! Each workload below is the Rust original of a table in tools/ref_twin/twin_workloads.py (same name); compare.py
//! runs the oracle on those tables and diffs the two JSONL streams.
//!
//! Build (outside this image — it has no rustc):  RUSTFLAGS="--cfg madsim" cargo run --release -- <workload|all> <seed0> <count> [loss]

use madsim::net::{Endpoint, NetSim};
use madsim::runtime::{Handle, Runtime};
use madsim::time::{self, Duration, Instant};
use std::net::SocketAddr;
use std::panic::{catch_unwind, AssertUnwindSafe};
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::{Arc, Mutex};

/// Values a workload makes observable, in execution order (the repo folds the same values into `obs_hash`).
#[derive(Clone, Default)]
struct Obs(Arc<Mutex<Vec<u64>>>);
impl Obs {
    fn push(&self, v: u64) { self.0.lock().unwrap().push(v); }
    fn take(&self) -> Vec<u64> { std::mem::take(&mut *self.0.lock().unwrap()) }
}

struct Tail { elapsed_ns: u64, msg_count: u64 }

/// The fingerprint tail every twin's main future ends with:
/// repo side = `trace_instant(); random_u32(); trace_val()` (tools/ref_twin/twin_workloads.py::fingerprint_tail).
fn fingerprint_tail(t0: Instant, obs: &Obs) -> Tail {
    let elapsed_ns = t0.elapsed().as_nanos() as u64;
    obs.push(elapsed_ns);                                   // MS_OP_TRACE_TIME a=1
    let msg_count = NetSim::current().stat().msg_count;
    let r: u32 = madsim::rand::random();                    // MS_OP_RANDOM a=0: one GlobalRng::with, next_u32
    obs.push(r as u64);                                     // MS_OP_TRACE_TIME a=2
    Tail { elapsed_ns, msg_count }
}

fn addr(node: usize, port: u16) -> SocketAddr { format!("10.0.0.{node}:{port}").parse().unwrap() }

const PING: &[u8; 4] = b"ping";
const PONG: &[u8; 4] = b"pong";

/// N nodes 10.0.0.i, pairs (1,2),(3,4)..; pinger = bind, sleep(1 s), R x {send ping, recv pong};
/// ponger = bind, R x {recv ping, send pong to `from`}; main spawns in node order and awaits the handles in order.
async fn pingpong(n_nodes: usize, rounds: usize, obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let mut handles = vec![];
    for i in 1..=n_nodes {
        let node = h.create_node().ip(addr(i, 1).ip()).build();
        let me = addr(i, 1);
        if i % 2 == 1 {
            let peer = addr(i + 1, 1);
            handles.push(node.spawn(async move {
                let ep = Endpoint::bind(me).await.unwrap();
                time::sleep(Duration::from_secs(1)).await;
                let mut buf = [0u8; 16];
                for _ in 0..rounds {
                    ep.send_to(peer, 1, PING).await.unwrap();
                    let (len, _from) = ep.recv_from(1, &mut buf).await.unwrap();
                    assert_eq!(&buf[..len], PONG);
                }
            }));
        } else {
            handles.push(node.spawn(async move {
                let ep = Endpoint::bind(me).await.unwrap();
                let mut buf = [0u8; 16];
                for _ in 0..rounds {
                    let (len, from) = ep.recv_from(1, &mut buf).await.unwrap();
                    assert_eq!(&buf[..len], PING);
                    ep.send_to(from, 1, PONG).await.unwrap();
                }
            }));
        }
    }
    for jh in handles { jh.await.unwrap(); }
    fingerprint_tail(t0, &obs)
}

/// SURVEY Appendix B minimal trace: `block_on(sleep(1 s))` (+ the tail).
async fn sleep_1s(obs: Obs) -> Tail {
    let t0 = Instant::now();
    time::sleep(Duration::from_secs(1)).await;
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:1018-1041 with sequential awaits: 3 tasks x 5 x { observe(i*10 + remaining); yield_now() }.
/// (The repo's loop counter counts DOWN: the observed value is i*10 + (5 - j), j = 0..5.)
async fn yield_order(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let mut tasks = vec![];
    for i in 0..3u64 {
        let obs = obs.clone();
        tasks.push(madsim::task::spawn(async move {
            for j in 0..5u64 {
                obs.push(i * 10 + (5 - j));
                tokio::task::yield_now().await;
            }
        }));
    }
    for t in tasks { t.await.unwrap(); }
    fingerprint_tail(t0, &obs)
}

/// Equal-deadline timers: 6 tasks all `sleep(10 ms)` from the same instant, each observing its index when it wakes —
/// pins the BinaryHeap tie order of naive-timer and the ready-queue draw together.
async fn timer_ties(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let mut tasks = vec![];
    for i in 0..6u64 {
        let obs = obs.clone();
        tasks.push(madsim::task::spawn(async move {
            for k in 0..3u64 {
                time::sleep(Duration::from_millis(10)).await;
                obs.push(i * 100 + k);
            }
        }));
    }
    for t in tasks { t.await.unwrap(); }
    fingerprint_tail(t0, &obs)
}

/// Deadlines equal to the nanosecond: five pairs of tasks looping `sleep(1 ms + 75 ns)` / `sleep(1 ms)` — polled back to back with a
/// poll cost of exactly 75 ns in between (task/mod.rs:319-321: 50..100 ns) the two timers tie, and the BinaryHeap's array order decides
/// who wakes first (about every seventh seed).  `timer_ties` above only gets deadlines 50..100 ns apart.
async fn ns_ties(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let mut tasks = vec![];
    for p in 0..5u64 {
        let o = obs.clone();
        tasks.push(madsim::task::spawn(async move {
            for _ in 0..40 {
                time::sleep(Duration::from_nanos(1_000_075)).await;
                o.push(0x100 + 2 * p);
            }
        }));
        let o = obs.clone();
        tasks.push(madsim::task::spawn(async move {
            for _ in 0..40 {
                time::sleep(Duration::from_millis(1)).await;
                o.push(0x101 + 2 * p);
            }
        }));
    }
    for t in tasks { t.await.unwrap(); }
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:859-897 `kill`.
async fn lifecycle_kill(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node1 = h.create_node().build();
    let node2 = h.create_node().build();
    let (flag1, flag2) = (Arc::new(AtomicUsize::new(0)), Arc::new(AtomicUsize::new(0)));
    let f = flag1.clone();
    node1.spawn(async move { loop { time::sleep(Duration::from_secs(2)).await; f.fetch_add(2, Ordering::Relaxed); } });
    let f = flag2.clone();
    node2.spawn(async move { loop { time::sleep(Duration::from_secs(2)).await; f.fetch_add(2, Ordering::Relaxed); } });
    time::sleep_until(t0 + Duration::from_secs(3)).await;
    assert_eq!(flag1.load(Ordering::Relaxed), 2);
    assert_eq!(flag2.load(Ordering::Relaxed), 2);
    h.kill(node1.id());
    h.kill(node1.id());
    assert!(h.is_exit(node1.id()));
    time::sleep_until(t0 + Duration::from_secs(5)).await;
    assert_eq!(flag1.load(Ordering::Relaxed), 2);
    assert_eq!(flag2.load(Ordering::Relaxed), 4);
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:899-936 `restart` (an init task; kill + restart; the flag restarts from 0).
async fn lifecycle_restart(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let flag = Arc::new(AtomicUsize::new(0));
    let flag_ = flag.clone();
    let node = h.create_node().init(move || {
        let flag = flag_.clone();
        async move {
            flag.store(0, Ordering::Relaxed);
            loop { time::sleep(Duration::from_secs(2)).await; flag.fetch_add(2, Ordering::Relaxed); }
        }
    }).build();
    time::sleep_until(t0 + Duration::from_secs(3)).await;
    assert_eq!(flag.load(Ordering::Relaxed), 2);
    h.kill(node.id());
    h.restart(node.id());
    assert!(!h.is_exit(node.id()));
    time::sleep_until(t0 + Duration::from_secs(6)).await;
    assert_eq!(flag.load(Ordering::Relaxed), 2);
    time::sleep_until(t0 + Duration::from_secs(8)).await;
    assert_eq!(flag.load(Ordering::Relaxed), 4);
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:938-962 `restart_on_panic`: three panics, restart delays drawn from 1..10 s (UniformDuration, Medium path).
async fn lifecycle_restart_on_panic(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let flag = Arc::new(AtomicUsize::new(0));
    let flag_ = flag.clone();
    h.create_node().init(move || {
        let flag = flag_.clone();
        async move { if flag.fetch_add(1, Ordering::Relaxed) < 3 { panic!(); } }
    }).restart_on_panic().build();
    time::sleep(Duration::from_secs(60)).await;
    assert_eq!(flag.load(Ordering::Relaxed), 4);
    fingerprint_tail(t0, &obs)
}

/// net/endpoint.rs:409-443 `receiver_drop` without the Barrier (whose wake order would add another dependency):
/// the sender waits 2 s instead, the receiver times out after 1 s, sleeps 2 s, and receives again.
async fn receiver_drop(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let (a1, a2) = (addr(1, 1), addr(2, 1));
    let node1 = h.create_node().ip(a1.ip()).build();
    let node2 = h.create_node().ip(a2.ip()).build();
    let s = node1.spawn(async move {
        let ep = Endpoint::bind(a1).await.unwrap();
        time::sleep(Duration::from_secs(2)).await;
        ep.send_to(a2, 1, &[1]).await.unwrap();
    });
    let r = node2.spawn(async move {
        let ep = Endpoint::bind(a2).await.unwrap();
        let mut buf = vec![0; 0x10];
        time::timeout(Duration::from_secs(1), ep.recv_from(1, &mut buf)).await.err().unwrap();
        let (len, from) = ep.recv_from(1, &mut buf).await.unwrap();
        assert_eq!((len, from), (1, a1));
    });
    s.await.unwrap();
    r.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// net/endpoint.rs:516-548 `localhost`, sleeps instead of the Barrier: 127.0.0.1-bound endpoints, a datagram to an address
/// nobody listens on (dropped after the draws), the receiver seeing the sender's real IP, a reply that finds no socket.
async fn localhost(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node1 = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
    let node2 = h.create_node().ip("10.0.0.2".parse().unwrap()).build();
    let f1 = node1.spawn(async move {
        let ep1 = Endpoint::bind("127.0.0.1:1").await.unwrap();
        let ep2 = Endpoint::bind("10.0.0.1:2").await.unwrap();
        time::timeout(Duration::from_secs(1), ep1.recv_from(1, &mut [])).await.expect_err("localhost endpoint should not receive from other nodes");
        let mut buf = [0u8; 4];
        let (_, from) = ep2.recv_from(1, &mut buf).await.unwrap();
        assert_eq!(from.to_string(), "10.0.0.2:1");
        ep2.send_to(from, 1, &[7]).await.unwrap();
    });
    let f2 = node2.spawn(async move {
        let ep = Endpoint::bind("127.0.0.1:1").await.unwrap();
        time::sleep(Duration::from_millis(5)).await;
        ep.send_to("10.0.0.1:1", 1, &[1]).await.unwrap();
        ep.send_to("10.0.0.1:2", 1, &[1]).await.unwrap();
        time::timeout(Duration::from_secs(2), ep.recv_from(1, &mut [])).await.expect_err("the reply went to 10.0.0.2:1");
    });
    f1.await.unwrap();
    f2.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:964-982 `restart_on_panic_matching`: panics "0" and "1" restart the node, "2" unwinds out of block_on.
async fn restart_on_panic_matching(_obs: Obs) -> Tail {
    let h = Handle::current();
    let flag = Arc::new(AtomicUsize::new(0));
    h.create_node().init(move || {
        let flag = flag.clone();
        async move { panic!("{}", flag.fetch_add(1, Ordering::Relaxed)); }
    }).restart_on_panic_matching("0").restart_on_panic_matching("1").build();
    time::sleep(Duration::from_secs(120)).await;
    unreachable!("the third panic ends the run")
}

/// net/endpoint.rs:470-513 `bind` (v4 cases) + network.rs:224-236 in detail; repo side: twin_workloads.py::bind_ephemeral.
async fn bind_ephemeral(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
    let _other = h.create_node().ip("10.0.0.2".parse().unwrap()).build();
    let o = obs.clone();
    let f = node.spawn(async move {
        let port = |ep: &Endpoint| ep.local_addr().unwrap().port() as u64;
        let any_a = Endpoint::bind("0.0.0.0:0").await.unwrap();
        o.push(port(&any_a));
        let lo_a = Endpoint::bind("127.0.0.1:0").await.unwrap();
        o.push(port(&lo_a));
        let err = Endpoint::bind("10.0.0.2:0").await.err().unwrap();
        assert_eq!(err.kind(), std::io::ErrorKind::AddrNotAvailable);
        let ip100 = Endpoint::bind("10.0.0.1:100").await.unwrap();
        o.push(port(&ip100));
        drop(ip100);
        let _ip100 = Endpoint::bind("10.0.0.1:100").await.unwrap();
        let any_b = Endpoint::bind("0.0.0.0:0").await.unwrap();
        o.push(port(&any_b));
        let _any3 = Endpoint::bind("0.0.0.0:3").await.unwrap();
        let any_c = Endpoint::bind("0.0.0.0:0").await.unwrap();
        o.push(port(&any_c));
        drop(any_a);
        drop(any_c);
        let any_c = Endpoint::bind("0.0.0.0:0").await.unwrap();
        o.push(port(&any_c));
        let any_a = Endpoint::bind("0.0.0.0:0").await.unwrap();
        o.push(port(&any_a));
        drop((lo_a, any_b, any_c, any_a));
    });
    f.await.unwrap();
    fingerprint_tail(t0, &obs)
}

fn payload(v: u32) -> Box<dyn std::any::Any + Send + Sync> { Box::new(v) }
fn value(p: Box<dyn std::any::Any + Send + Sync>) -> u64 { *p.downcast::<u32>().unwrap() as u64 }

/// connect1 through general addresses (net/mod.rs:337-364): a 0.0.0.0:2379 listener, a client on 0.0.0.0:0 dialling
/// 10.0.0.1:2379, three request / response pairs, then the server side is gone; repo side: twin_workloads.py::channel_wildcard.
async fn channel_wildcard(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let ns = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
    let nc = h.create_node().ip("10.0.0.2".parse().unwrap()).build();
    let o = obs.clone();
    let _srv = ns.spawn(async move {
        let ep = Endpoint::bind("0.0.0.0:2379").await.unwrap();
        let (tx, mut rx, _) = ep.accept1().await.unwrap();
        for _ in 0..3 {
            o.push(value(rx.recv().await.unwrap()));
            tx.send(payload(0x22)).await.unwrap();
        }
    });
    let cl = nc.spawn(async move {
        let ep = Endpoint::bind("0.0.0.0:0").await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        let (tx, mut rx) = ep.connect1("10.0.0.1:2379".parse().unwrap()).await.unwrap();
        for _ in 0..3 {
            tx.send(payload(0x11)).await.unwrap();
            assert_eq!(value(rx.recv().await.unwrap()), 0x22);
        }
        assert_eq!(rx.recv().await.err().unwrap().kind(), std::io::ErrorKind::ConnectionReset);
    });
    cl.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// Sender / Receiver keep their Endpoint's Arc<BindGuard> (endpoint.rs:181-210): the listener is dropped while the connection
/// it accepted lives on in a handler task; binding the address again fails until the handler is done.  The client's second
/// connect1 succeeds (the address is in the table) but nobody can accept it: its first recv is a ConnectionReset.
/// repo side: twin_workloads.py::guard_keeps_address.
async fn guard_keeps_address(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let ns = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
    let nc = h.create_node().ip("10.0.0.2".parse().unwrap()).build();
    let o = obs.clone();
    let srv = ns.spawn(async move {
        let ep = Endpoint::bind("10.0.0.1:7").await.unwrap();
        let (tx, mut rx, _) = ep.accept1().await.unwrap();
        let handler = madsim::task::spawn(async move {
            assert_eq!(value(rx.recv().await.unwrap()), 1);
            time::sleep(Duration::from_millis(50)).await;
            tx.send(payload(2)).await.unwrap();
        });
        drop(ep);
        time::sleep(Duration::from_millis(5)).await;
        o.push(match Endpoint::bind("10.0.0.1:7").await { Ok(_) => 0, Err(e) => { assert_eq!(e.kind(), std::io::ErrorKind::AddrInUse); 1 } });
        handler.await.unwrap();
        time::sleep(Duration::from_millis(5)).await;
        o.push(match Endpoint::bind("10.0.0.1:7").await { Ok(_) => 0, Err(_) => 1 });
    });
    let o = obs.clone();
    let cl = nc.spawn(async move {
        let ep = Endpoint::bind("10.0.0.2:1").await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        let (tx, mut rx) = ep.connect1("10.0.0.1:7".parse().unwrap()).await.unwrap();
        tx.send(payload(1)).await.unwrap();
        time::sleep(Duration::from_millis(20)).await;
        let o2 = o.clone();
        let second = madsim::task::spawn(async move {     // (a task of its own, as in the table form: one (tx, rx) pair per task)
            let ep2 = Endpoint::bind("10.0.0.2:2").await.unwrap();
            let (_tx2, mut rx2) = ep2.connect1("10.0.0.1:7".parse().unwrap()).await.unwrap();
            o2.push(match rx2.recv().await { Ok(_) => 0, Err(e) => { assert_eq!(e.kind(), std::io::ErrorKind::ConnectionReset); 1 } });
        });
        second.await.unwrap();
        o.push(value(rx.recv().await.unwrap()));
    });
    srv.await.unwrap();
    cl.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// A value whose Drop runs a closure: `impl Drop for A { fn drop(&mut self) { spawn(..) } }` of task/mod.rs:1188-1196.
struct OnDrop<F: FnOnce() + Send + 'static>(Option<F>);
impl<F: FnOnce() + Send + 'static> Drop for OnDrop<F> {
    fn drop(&mut self) { if let Some(f) = self.0.take() { f() } }
}

/// task/mod.rs:1184-1216 `spawn_in_future_drop_by_aborting_task`: the task spawned in A::drop runs, on the aborted task's node.
async fn spawn_in_drop_abort(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node = h.create_node().build();
    let node_id = node.id();
    let ran_on = Arc::new(Mutex::new(None));
    let r = ran_on.clone();
    let a = OnDrop(Some(move || { madsim::task::spawn(async move { *r.lock().unwrap() = Some(madsim::plugin::node()); }); }));
    let jh = node.spawn(async move { drop(a) });
    jh.abort();
    assert!(jh.await.unwrap_err().is_cancelled());
    time::sleep(Duration::from_secs(57257)).await;
    time::sleep(Duration::from_secs(57257)).await;
    assert_eq!(*ran_on.lock().unwrap(), Some(node_id));
    obs.push(1);
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:1219-1253 `spawn_in_future_drop_by_killing_node`: spawning on the killed node succeeds, the task never runs.
async fn spawn_in_drop_kill(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node = h.create_node().build();
    let dropped = Arc::new(AtomicUsize::new(0));
    let d = dropped.clone();
    let a = OnDrop(Some(move || { madsim::task::spawn(async move { unreachable!() }); d.store(1, Ordering::Relaxed); }));
    let jh = node.spawn(async move { drop(a) });
    h.kill(node.id());
    assert!(jh.await.unwrap_err().is_cancelled());
    assert_eq!(dropped.load(Ordering::Relaxed), 1);
    time::sleep(Duration::from_secs(57257)).await;
    time::sleep(Duration::from_secs(57257)).await;
    obs.push(dropped.load(Ordering::Relaxed) as u64);
    fingerprint_tail(t0, &obs)
}

/// task::spawn from a task that has just restarted its own node: Spawner::current() is the caller's own Arc<NodeInfo>
/// (task/mod.rs:592-599), so the spawned task belongs to the dead incarnation and never runs; obs <- the three counters.
async fn spawn_after_own_restart(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let (child_ran, saboteur, inits) = (Arc::new(AtomicUsize::new(0)), Arc::new(AtomicUsize::new(0)), Arc::new(AtomicUsize::new(0)));
    let i = inits.clone();
    let node = h.create_node().init(move || {
        let i = i.clone();
        async move { i.fetch_add(1, Ordering::Relaxed); time::sleep(Duration::from_secs(10)).await; }
    }).build();
    time::sleep(Duration::from_millis(5)).await;
    let id = node.id();
    let (c, s) = (child_ran.clone(), saboteur.clone());
    node.spawn(async move {
        Handle::current().restart(id);
        madsim::task::spawn(async move { c.fetch_add(1, Ordering::Relaxed); });
        s.fetch_add(1, Ordering::Relaxed);
        time::sleep(Duration::from_millis(1)).await;
        s.fetch_add(10, Ordering::Relaxed);
    });
    time::sleep(Duration::from_secs(1)).await;
    obs.push(child_ran.load(Ordering::Relaxed) as u64);
    obs.push(saboteur.load(Ordering::Relaxed) as u64);
    obs.push(inits.load(Ordering::Relaxed) as u64);
    fingerprint_tail(t0, &obs)
}

/// `handle.await` moves the JoinHandle into the await: the awaiter gets the first worker's outcome (Ok after 5 ms) although the
/// same closure was spawned again meanwhile into the variable it came from, and that second worker aborted.
async fn join_names_its_task(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node = h.create_node().build();
    let done = Arc::new(AtomicUsize::new(0));
    let worker = {
        let done = done.clone();
        move || { let done = done.clone(); async move { time::sleep(Duration::from_millis(5)).await; done.fetch_add(1, Ordering::Relaxed); } }
    };
    let slot: Arc<Mutex<Option<madsim::task::JoinHandle<()>>>> = Arc::new(Mutex::new(None));
    let first = node.spawn(worker());
    let (slot2, worker2) = (slot.clone(), worker.clone());
    node.spawn(async move {
        time::sleep(Duration::from_millis(1)).await;
        let second = madsim::task::spawn(worker2());
        time::sleep(Duration::from_millis(1)).await;
        second.abort();
        *slot2.lock().unwrap() = Some(second);
        time::sleep(Duration::from_millis(20)).await;
    });
    first.await.unwrap();
    let waited = t0.elapsed();
    assert!(waited >= Duration::from_millis(5) && waited < Duration::from_millis(7));
    obs.push(done.load(Ordering::Relaxed) as u64);
    let second = slot.lock().unwrap().take().unwrap();
    assert!(second.await.unwrap_err().is_cancelled());
    fingerprint_tail(t0, &obs)
}

/// A task that aborts its own JoinHandle keeps running until it yields, then is dropped: obs <- 1 (the work before its next
/// await happened, the work behind it did not).
async fn abort_own_handle(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node = h.create_node().build();
    let flag = Arc::new(AtomicUsize::new(0));
    let own: Arc<Mutex<Option<madsim::task::AbortHandle>>> = Arc::new(Mutex::new(None));
    let (f, o) = (flag.clone(), own.clone());
    let jh = node.spawn(async move {
        time::sleep(Duration::from_millis(1)).await;
        o.lock().unwrap().as_ref().unwrap().abort();
        f.fetch_add(1, Ordering::Relaxed);
        time::sleep(Duration::from_millis(5)).await;
        f.fetch_add(10, Ordering::Relaxed);
    });
    *own.lock().unwrap() = Some(jh.abort_handle());
    assert!(jh.await.unwrap_err().is_cancelled());
    time::sleep(Duration::from_millis(20)).await;
    obs.push(flag.load(Ordering::Relaxed) as u64);
    fingerprint_tail(t0, &obs)
}

// ---- round 3: rpc hooks, substring panic patterns, a datagram in flight across a re-bind, the DSL-built ping-pong ------------

#[derive(madsim::net::rpc::Serialize, madsim::net::rpc::Deserialize, madsim::Request)]
#[rtype("u32")]
struct Echo(u32);
#[derive(madsim::net::rpc::Serialize, madsim::net::rpc::Deserialize, madsim::Request)]
#[rtype("u32")]
struct Other(u32);

/// NetSim::hook_rpc_req / hook_rpc_rsp (net/mod.rs:240-284; consulted by NetSim::send, :307-311 and :321-328).
/// Table: twin_workloads.py::rpc_hooks (the handler's `obs.push(1)` is its MS_OP_TRACE 1).
async fn rpc_hooks(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let net = NetSim::current();
    let ns = h.create_node().ip(addr(1, 1).ip()).build();
    let (n1, n2, n3) = (h.create_node().ip(addr(2, 1).ip()).build(), h.create_node().ip(addr(3, 1).ip()).build(),
                        h.create_node().ip(addr(4, 1).ip()).build());
    let asv = addr(1, 1);
    net.hook_rpc_req(n1.id(), |req: &Echo| req.0 != 5);                 // m.hook_rpc_req(n1, 0, code=5)
    net.hook_rpc_rsp(n2.id(), |_rsp: &u32| false);                      // m.hook_rpc_rsp(n2): every response is dropped
    let o = obs.clone();
    let _srv = ns.spawn(async move {
        let ep = Endpoint::bind(asv).await.unwrap();
        ep.add_rpc_handler(move |_req: Echo| { o.push(1); async move { 42u32 } });
        std::future::pending::<()>().await;
    });
    let timeout = Duration::from_millis(100);
    let c1 = n1.spawn(async move {
        let ep = Endpoint::bind(addr(2, 1)).await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        assert_eq!(ep.call_timeout(asv, Echo(5), timeout).await.unwrap_err().kind(), std::io::ErrorKind::TimedOut);
        assert_eq!(ep.call_timeout(asv, Echo(6), timeout).await.unwrap(), 42);
    });
    let c2 = n2.spawn(async move {
        let ep = Endpoint::bind(addr(3, 1)).await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        assert_eq!(ep.call_timeout(asv, Echo(7), timeout).await.unwrap_err().kind(), std::io::ErrorKind::TimedOut);
        time::sleep(Duration::from_millis(400)).await;
        assert_eq!(ep.call_timeout(asv, Echo(8), timeout).await.unwrap(), 42);
    });
    let c3 = n3.spawn(async move {
        let ep = Endpoint::bind(addr(4, 1)).await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        for _ in 0..3 { assert_eq!(ep.call(asv, Echo(9)).await.unwrap(), 42); }
    });
    time::sleep(Duration::from_millis(300)).await;
    net.hook_rpc_rsp(n2.id(), |rsp: &u32| *rsp != 99);                  // m.hook_rpc_rsp(n2, code=99): replaces the hook
    c1.await.unwrap(); c2.await.unwrap(); c3.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// task/mod.rs:297-300 `error_msg.contains(pattern)` with literal messages.  Table: twin_workloads.py::panic_substrings.
/// Ends in the panic "out of memory" at 100 s (verdict "panic", like the #[should_panic] test it generalises).
async fn panic_substrings(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let (fa, fb) = (Arc::new(AtomicUsize::new(0)), Arc::new(AtomicUsize::new(0)));
    let f = fa.clone();
    // two init tasks on node A: NodeBuilder::init takes ONE closure, so the node's init spawns the second body beside the first
    let _a = h.create_node().restart_on_panic_matching("disk").restart_on_panic_matching("net").init(move || {
        let f = f.clone();
        async move {
            let second = madsim::task::spawn(async move { time::sleep(Duration::from_secs(7)).await; panic!("network reset"); });
            f.fetch_add(1, Ordering::Relaxed);
            time::sleep(Duration::from_secs(3)).await;
            let _keep = second;
            panic!("disk full");
        }
    }).build();
    let f = fb.clone();
    let _b = h.create_node().restart_on_panic_matching("reset").init(move || {
        let f = f.clone();
        async move { f.fetch_add(1, Ordering::Relaxed); time::sleep(Duration::from_secs(20)).await; panic!("network reset"); }
    }).build();
    let _c = h.create_node().restart_on_panic_matching("timeout").init(|| async {
        time::sleep(Duration::from_secs(100)).await; panic!("out of memory");
    }).build();
    time::sleep(Duration::from_secs(50)).await;
    assert!(fa.load(Ordering::Relaxed) >= 3 && fb.load(Ordering::Relaxed) >= 2);
    obs.push(1);
    time::sleep(Duration::from_secs(100)).await;
    fingerprint_tail(t0, &obs)
}

/// A datagram in flight to a socket that is closed and re-bound before it lands: the delivery closure holds the OLD
/// `Arc<dyn Socket>` (net/mod.rs:318-330).  Table: twin_workloads.py::rebind_in_flight; 0xFFFF_FFFF = Err(Elapsed).
async fn rebind_in_flight(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let (n1, n2) = (h.create_node().ip(addr(1, 1).ip()).build(), h.create_node().ip(addr(2, 1).ip()).build());
    let o = obs.clone();
    let rx = n2.spawn(async move {
        let ep = Endpoint::bind(addr(2, 1)).await.unwrap();
        time::sleep(Duration::from_millis(14)).await;
        drop(ep);
        let ep = Endpoint::bind(addr(2, 1)).await.unwrap();
        let mut buf = [0u8; 16];
        for (tag, ms) in [(1u64, 200u64), (2, 400)] {
            match time::timeout(Duration::from_millis(ms), ep.recv_from(tag, &mut buf)).await {
                Ok(r) => { let (len, _) = r.unwrap(); let mut w = [0u8; 4]; w[..len].copy_from_slice(&buf[..len]); o.push(u32::from_le_bytes(w) as u64) }
                Err(_) => o.push(0xFFFF_FFFF),
            }
        }
    });
    let tx = n1.spawn(async move {
        let ep = Endpoint::bind(addr(1, 1)).await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        ep.send_to(addr(2, 1), 1, &5u32.to_le_bytes()).await.unwrap();
        time::sleep(Duration::from_millis(100)).await;
        ep.send_to(addr(2, 1), 2, &7u32.to_le_bytes()).await.unwrap();
    });
    rx.await.unwrap(); tx.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// IpVirtualServer changed while datagrams flow (net/ipvs.rs:50-105: add_service / add_server / del_server / del_service from a
/// task; the wrap of an index left beyond a shortened list; retain; insert over an existing service).
/// Table: twin_workloads.py::ipvs_runtime.  0xFFFF_FFFF = Err(Elapsed).
async fn ipvs_runtime(obs: Obs) -> Tail {
    use madsim::net::ipvs::{Scheduler, ServiceAddr};
    let t0 = Instant::now();
    let h = Handle::current();
    let nodes: Vec<_> = (1..=4u8).map(|i| h.create_node().ip(addr(i, 1).ip()).build()).collect();
    for (i, code) in [(1u8, 0xAu32), (2, 0xB), (3, 0xC)] {
        nodes[i as usize - 1].spawn(async move {
            let ep = Endpoint::bind(addr(i, 1)).await.unwrap();
            let mut buf = [0u8; 16];
            while let Ok(r) = time::timeout(Duration::from_millis(400), ep.recv_from(1, &mut buf)).await {
                let (_, from) = r.unwrap();
                ep.send_to(from, 2, &code.to_le_bytes()).await.unwrap();
            }
        });
    }
    let o = obs.clone();
    let op = nodes[3].spawn(async move {
        let ep = Endpoint::bind(addr(4, 1)).await.unwrap();
        time::sleep(Duration::from_millis(5)).await;
        let ipvs = NetSim::current().global_ipvs();
        let svc = || ServiceAddr::Tcp("1.1.1.1:80".into());
        let vip: SocketAddr = "1.1.1.1:80".parse().unwrap();
        let (a, b, c) = ("10.0.0.1:1", "10.0.0.2:1", "10.0.0.3:1");
        let _ = b;
        macro_rules! probe { ($n:expr) => { for _ in 0..$n {
            ep.send_to(vip, 1, &7u32.to_le_bytes()).await.unwrap();
            let mut buf = [0u8; 16];
            match time::timeout(Duration::from_millis(30), ep.recv_from(2, &mut buf)).await {
                Ok(r) => { let (len, _) = r.unwrap(); let mut w = [0u8; 4]; w[..len].copy_from_slice(&buf[..len]); o.push(u32::from_le_bytes(w) as u64) }
                Err(_) => o.push(0xFFFF_FFFF),
            }
        } } }
        ipvs.add_service(svc(), Scheduler::RoundRobin); probe!(1);
        ipvs.add_server(svc(), a); ipvs.add_server(svc(), "10.0.0.2:1"); ipvs.add_server(svc(), c); probe!(3);
        ipvs.del_server(svc(), c); probe!(1);
        ipvs.add_server(svc(), a); ipvs.add_server(svc(), c); probe!(3);
        ipvs.del_server(svc(), a); probe!(1);
        ipvs.del_service(svc()); probe!(1);
        ipvs.add_service(svc(), Scheduler::RoundRobin); probe!(1);
        ipvs.add_server(svc(), c); probe!(2);
    });
    op.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// IpVirtualServer (net/ipvs.rs) consulted by NetSim::send and connect1 (net/mod.rs:312-317,345-350).
/// Table: twin_workloads.py::ipvs_round_robin.  0xFFFF_FFFF = Err(Elapsed).
async fn ipvs_round_robin(obs: Obs) -> Tail {
    use madsim::net::ipvs::{Scheduler, ServiceAddr};
    let t0 = Instant::now();
    let h = Handle::current();
    let ipvs = NetSim::current().global_ipvs();
    ipvs.add_service(ServiceAddr::Tcp("1.1.1.1:80".into()), Scheduler::RoundRobin);
    ipvs.add_server(ServiceAddr::Tcp("1.1.1.1:80".into()), "10.0.0.1:1");
    ipvs.add_server(ServiceAddr::Tcp("1.1.1.1:80".into()), "10.0.0.2:1");
    let nodes = [h.create_node().ip(addr(1, 1).ip()).build(), h.create_node().ip(addr(2, 1).ip()).build(),
                 h.create_node().ip(addr(3, 1).ip()).build()];
    let mut servers = vec![];
    for i in 0..2 {
        let o = obs.clone();
        servers.push(nodes[i].spawn(async move {
            let ep = Endpoint::bind("0.0.0.0:1").await.unwrap();
            let (_tx, mut rx, _) = ep.accept1().await.unwrap();
            o.push(value(rx.recv().await.unwrap()));
            let mut buf = [0u8; 16];
            for _ in 0..2 {
                match time::timeout(Duration::from_millis(300), ep.recv_from(1, &mut buf)).await {
                    Ok(r) => { let (len, _) = r.unwrap(); let mut w = [0u8; 4]; w[..len].copy_from_slice(&buf[..len]); o.push(u32::from_le_bytes(w) as u64) }
                    Err(_) => o.push(0xFFFF_FFFF),
                }
            }
        }));
    }
    let f3 = nodes[2].spawn(async move {
        time::sleep(Duration::from_millis(50)).await;
        let ep = Endpoint::bind(addr(3, 7)).await.unwrap();
        let vip: SocketAddr = "1.1.1.1:80".parse().unwrap();
        let mut held = None;                                         // the table's task holds one (tx, rx) pair at a time: its
        for v in [1u32, 2] {                                         // second connect1 lets go of the first (nobody is parked on it)
            drop(held.take());
            let (tx, rx) = ep.connect1(vip).await.unwrap();
            tx.send(payload(v)).await.unwrap();
            held = Some((tx, rx));
            time::sleep(Duration::from_millis(30)).await;
        }
        for k in 0..3u32 {
            ep.send_to(vip, 1, &(10 + k).to_le_bytes()).await.unwrap();
            time::sleep(Duration::from_millis(20)).await;
        }
    });
    for s in servers { s.await.unwrap(); }
    f3.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// `NetSim::update_config(|c| c.send_latency = ..)` between datagrams (net/mod.rs:138-141 -> network.rs:129; sampled at :267): 100..101 ms,
/// 1.5..3.5 s, 1..2 ns; the receiver observes when each datagram arrives.  Table: twin_workloads.py::update_config_latency.
async fn update_config_latency(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node1 = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
    let node2 = h.create_node().ip("10.0.0.2".parse().unwrap()).build();
    let o = obs.clone();
    let rx = node2.spawn(async move {
        let ep = Endpoint::bind("10.0.0.2:1").await.unwrap();
        let mut buf = [0u8; 4];
        for tag in 1..=4u64 {
            ep.recv_from(tag, &mut buf).await.unwrap();
            o.push(t0.elapsed().as_nanos() as u64);                 // MS_OP_TRACE_TIME a=1
        }
    });
    let tx = node1.spawn(async move {
        let t1 = Instant::now();                                    // the table's `mark`
        let ep = Endpoint::bind("10.0.0.1:1").await.unwrap();
        time::sleep(Duration::from_millis(10)).await;
        let net = NetSim::current();
        ep.send_to("10.0.0.2:1", 1, &[0xA]).await.unwrap();
        net.update_config(|c| c.send_latency = Duration::from_millis(100)..Duration::from_millis(101));
        ep.send_to("10.0.0.2:1", 2, &[0xB]).await.unwrap();
        net.update_config(|c| c.send_latency = Duration::from_millis(1500)..Duration::from_millis(3500));
        ep.send_to("10.0.0.2:1", 3, &[0xC]).await.unwrap();
        time::sleep_until(t1 + Duration::from_secs(5)).await;
        net.update_config(|c| c.send_latency = Duration::from_nanos(1)..Duration::from_nanos(2));
        ep.send_to("10.0.0.2:1", 4, &[0xD]).await.unwrap();
    });
    rx.await.unwrap();
    tx.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// One task on both ends of its own connection (net/mod.rs:337-364, endpoint.rs:196-212): the pair `accept1` returns replaces the client
/// pair, whose handles drop at the assignment.  Table: twin_workloads.py::self_connect_accept.
async fn self_connect_accept(obs: Obs) -> Tail {
    let t0 = Instant::now();
    let h = Handle::current();
    let node = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
    let o = obs.clone();
    let f = node.spawn(async move {
        let ep = Endpoint::bind("10.0.0.1:1").await.unwrap();
        let (mut tx, mut rx) = ep.connect1("10.0.0.1:1".parse().unwrap()).await.unwrap();
        tx.send(payload(5)).await.unwrap();
        time::sleep(Duration::from_millis(20)).await;
        (tx, rx, _) = ep.accept1().await.unwrap();                  // the right-hand side first, then the old pair drops
        o.push(value(rx.recv().await.unwrap()));
        assert_eq!(rx.recv().await.err().unwrap().kind(), std::io::ErrorKind::ConnectionReset);
        o.push(1);
        drop(tx);
    });
    f.await.unwrap();
    fingerprint_tail(t0, &obs)
}

/// The 4-node ping-pong built ONCE with the Rust workload DSL (bindings/rust/madsim-hip, `pingpong_twin`) and interpreted on
/// real madsim by `madsim_hip::interp` — the table `Builder::run_workload` hands to the GPU runner, run here by the reference.
async fn pingpong4_dsl(obs: Obs) -> Tail {
    let w = madsim_hip::pingpong_twin(4, 64);
    let seen = madsim_hip::interp::main_future(&w, Vec::new()).await;
    for v in &seen.obs { obs.push(*v); }
    Tail { elapsed_ns: seen.obs[seen.obs.len() - 2], msg_count: seen.msg_count }
}

fn run_one(name: &str, seed: u64, loss: f64) -> String {
    let mut config = madsim::Config::default();
    config.net.packet_loss_rate = loss;
    let obs = Obs::default();
    let o = obs.clone();
    let name_owned = name.to_string();
    let mut log_hex = String::new();
    let res = catch_unwind(AssertUnwindSafe(|| {
        let rt = Runtime::with_seed_and_config(seed, config);
        #[cfg(feature = "rng-log")]
        rt.rng().enable_log();
        let tail = rt.block_on(async move {
            match name_owned.as_str() {
                "pingpong2" => pingpong(2, 64, o).await,
                "pingpong4" => pingpong(4, 64, o).await,
                "pingpong16" => pingpong(16, 8, o).await,
                "sleep_1s" => sleep_1s(o).await,
                "yield_order" => yield_order(o).await,
                "timer_ties" => timer_ties(o).await,
                "kill" => lifecycle_kill(o).await,
                "restart" => lifecycle_restart(o).await,
                "restart_on_panic" => lifecycle_restart_on_panic(o).await,
                "receiver_drop" => receiver_drop(o).await,
                "localhost" => localhost(o).await,
                "restart_on_panic_matching" => restart_on_panic_matching(o).await,
                "bind_ephemeral" => bind_ephemeral(o).await,
                "channel_wildcard" => channel_wildcard(o).await,
                "guard_keeps_address" => guard_keeps_address(o).await,
                "spawn_in_drop_abort" => spawn_in_drop_abort(o).await,
                "spawn_in_drop_kill" => spawn_in_drop_kill(o).await,
                "spawn_after_own_restart" => spawn_after_own_restart(o).await,
                "join_names_its_task" => join_names_its_task(o).await,
                "abort_own_handle" => abort_own_handle(o).await,
                "rpc_hooks" => rpc_hooks(o).await,
                "panic_substrings" => panic_substrings(o).await,
                "rebind_in_flight" => rebind_in_flight(o).await,
                "ipvs_round_robin" => ipvs_round_robin(o).await,
                "ipvs_runtime" => ipvs_runtime(o).await,
                "pingpong4_dsl" => pingpong4_dsl(o).await,
                "ns_ties" => ns_ties(o).await,
                "update_config_latency" => update_config_latency(o).await,
                "self_connect_accept" => self_connect_accept(o).await,
                other => panic!("unknown workload {other}"),
            }
        });
        #[cfg(feature = "rng-log")]
        { log_hex = rt.rng().take_log().unwrap().into_bytes().iter().map(|b| format!("{b:02x}")).collect(); }
        tail
    }));
    let obs_list = obs.take().iter().map(|v| v.to_string()).collect::<Vec<_>>().join(",");
    let log_field = if log_hex.is_empty() { String::new() } else { format!(",\"log_hex\":\"{log_hex}\"") };
    match res {
        Ok(t) => format!("{{\"workload\":\"{name}\",\"seed\":{seed},\"loss\":{loss},\"verdict\":\"pass\",\"elapsed_ns\":{},\"msg_count\":{},\"obs\":[{obs_list}]{log_field}}}",
                         t.elapsed_ns, t.msg_count),
        Err(e) => {
            let msg = e.downcast_ref::<String>().cloned().or_else(|| e.downcast_ref::<&str>().map(|s| s.to_string())).unwrap_or_default();
            let verdict = if msg.contains("all tasks will block forever") { "deadlock" }
                          else if msg.contains("time limit exceeded") { "time-limit" } else { "panic" };
            format!("{{\"workload\":\"{name}\",\"seed\":{seed},\"loss\":{loss},\"verdict\":\"{verdict}\",\"elapsed_ns\":null,\"msg_count\":null,\"obs\":[{obs_list}]}}")
        }
    }
}

const ALL: &[&str] = &["pingpong2", "pingpong4", "pingpong16", "sleep_1s", "yield_order", "timer_ties", "kill", "restart",
                       "restart_on_panic", "receiver_drop", "localhost", "restart_on_panic_matching", "bind_ephemeral",
                       "channel_wildcard", "guard_keeps_address", "spawn_in_drop_abort", "spawn_in_drop_kill",
                       "spawn_after_own_restart", "join_names_its_task", "abort_own_handle", "rpc_hooks", "panic_substrings",
                       "rebind_in_flight", "ipvs_round_robin", "ipvs_runtime", "pingpong4_dsl", "ns_ties", "update_config_latency",
                       "self_connect_accept"];

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() < 4 {
        eprintln!("usage: madsim-ref-twin <workload|all> <seed0> <count> [packet_loss_rate]\nworkloads: {}", ALL.join(" "));
        std::process::exit(2);
    }
    let (seed0, count): (u64, u64) = (args[2].parse().unwrap(), args[3].parse().unwrap());
    let loss: f64 = args.get(4).map(|s| s.parse().unwrap()).unwrap_or(0.0);
    std::panic::set_hook(Box::new(|_| {}));               // verdicts are data here, not noise on stderr
    let names: Vec<&str> = if args[1] == "all" { ALL.to_vec() } else { vec![args[1].as_str()] };
    for name in names {
        for seed in seed0..seed0 + count {
            // one OS thread per seed like Builder::run (builder.rs:134): madsim's context is thread-local
            let n = name.to_string();
            let line = std::thread::spawn(move || run_one(&n, seed, loss)).join().unwrap();
            println!("{line}");
        }
    }
}
