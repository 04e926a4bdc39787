/*
 * madsim_hip.h — C ABI of the MI355X many-seed runner for madsim's deterministic
 * discrete-event executor (Executor + TimeRuntime + GlobalRng + NetSim delivery queue).
 *
 * This is the drop-in boundary.  The reference has no FFI for its executor; the
 * seam this library plugs into is the per-seed fan-out of
 *     madsim::runtime::Builder::run            madsim/src/sim/runtime/builder.rs:121-162
 * (reached from #[madsim::test] / #[madsim::main], madsim-macros/src/lib.rs:134-151) and, per seed,
 *     Runtime::with_seed_and_config + block_on madsim/src/sim/runtime/mod.rs:53-69,127-130
 *     Executor::block_on / run_all_ready       madsim/src/sim/task/mod.rs:220-323
 * One call to madsim_hip_run_batch replaces `count` iterations of that fan-out: seeds
 * seed0 .. seed0+count-1, one GPU lane per seed.  INTEGRATION.md shows the Rust `extern "C"` stub.
 *
 * A GPU lane cannot poll an opaque Rust future (task/mod.rs:279-283), so the test body is handed
 * over as a *workload*: a table-driven actor program whose instructions are one-to-one with the
 * reference API calls a madsim test makes (Endpoint::bind/send_to/recv_from, time::sleep,
 * spawn/JoinHandle, Handle::kill/restart/pause/resume, NetSim::clog_*).  The executor semantics
 * underneath — ready-queue draw, 50..100 ns poll cost, timer heap order, 1 ms sleep floor, link
 * test and latency draw, mailbox tag matching, async-task wake rules — are restated bit-exactly.
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 */
#ifndef MADSIM_HIP_H
#define MADSIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADSIM_HIP_ABI_VERSION 4u

/* ------------------------------------------------------------------------------------------------
 * Workload: the actor program (read-only, caller-owned POD).
 * ---------------------------------------------------------------------------------------------- */

/* One instruction = 8 bytes.  Durations are encoded as  b seconds + imm nanoseconds. */
typedef struct madsim_insn {
    uint8_t  op;   /* enum madsim_op */
    uint8_t  a;
    uint16_t b;
    uint32_t imm;
} madsim_insn_t;

/* Each op stands for one reference API call; "await" ops may return Pending to the executor. */
enum madsim_op {
    /* -- task / control -- */
    MS_OP_DONE = 0,        /* async block returns: locals drop (endpoints close, net/mod.rs:483-493),
                              JoinHandle awaiter is woken (async-task)                  */
    MS_OP_SPAWN = 1,       /* a=prog: NodeHandle::spawn / task::spawn  (task/mod.rs:607-654);
                              handle[prog] := new task.  b&2: `async move` takes the spawner's (tx, rx) */
    MS_OP_JOIN = 2,        /* a=prog: handle[prog].await (task/join.rs:59-72). b&1: expect Err
                              (unwrap_err) else unwrap -> a mismatch panics the polling task         */
    MS_OP_ABORT = 3,       /* a=prog: handle[prog].abort() (task/join.rs:158-163)                    */
    MS_OP_YIELD = 4,       /* tokio::task::yield_now().await (re-export task/mod.rs:30)              */
    MS_OP_PANIC = 5,       /* a=0: panic!() with message code imm (0..254); a=1: panic!("{}", flag[b & 3] + imm): the message is
                              the decimal text of that value, which is also its code — values above
                              madsim_workload_t.panic_dyn_max yield MADSIM_UNSUPPORTED.  The code is what
                              NodeBuilder::restart_on_panic_matching looks at (task/mod.rs:297-300; madsim_workload_t.panic_match);
                              any other panic (failed assert, unwrap of an Err) carries code 255, which no pattern names */
    MS_OP_SET = 6,         /* a=reg(0..1): cnt[a] = imm (a loop bound; registers are 16 bit)          */
    MS_OP_DJNZ = 7,        /* a=reg, b=target: if (--cnt[a] != 0) goto b                              */
    MS_OP_JMP = 8,         /* b=target                                                               */
    MS_OP_TRACE = 9,       /* obs_hash <- fold(imm + (b&1 ? cnt[a] : 0)): an observable side effect
                              (the std::sync::mpsc send in task/mod.rs:1029)                         */
    MS_OP_BUILD = 15,      /* a=node: create_node().init(..).build() inside a task body: spawns the node's
                              MADSIM_PROG_INIT program(s) now (task/mod.rs:472-474)                  */
    /* -- time -- */
    MS_OP_SLEEP = 10,      /* time::sleep(b s + imm ns).await  (time/sleep.rs:5-8, mod.rs:111-124)   */
    MS_OP_MARK = 11,       /* t0 = Instant::now().  A program's SLEEP_UNTIL / ASSERT_ELAPSED must have a MARK at a lower pc: t0
                              is a local of the task body, and a use before its assignment is refused (MADSIM_E_WORKLOAD) as Rust would */
    MS_OP_SLEEP_UNTIL = 12,/* time::sleep_until(t0 + b s + imm ns).await (time/mod.rs:118-124)       */
    MS_OP_ASSERT_ELAPSED = 13, /* a=cmp (0 ==, 1 >=, 2 <): assert!(t0.elapsed() cmp b s + imm ns)    */
    MS_OP_ADVANCE = 14,    /* time::advance(b s + imm ns) (time/mod.rs:195-198, 103-106)             */
    /* -- net (datagram Endpoint API, net/endpoint.rs) -- */
    MS_OP_BIND = 20,       /* a=sock: Endpoint::bind(addr(sock)).await.unwrap() (endpoint.rs:23-36, network.rs:206-251).
                              b&1: without the unwrap — val := 0, MADSIM_VAL_ADDR_NOT_AVAILABLE or MADSIM_VAL_ADDR_IN_USE;
                              b&2: on success val := ep.local_addr().unwrap().port() (what a port-0 entry was given)        */
    MS_OP_SEND = 21,       /* a=src sock, b=(tag<<8)|dst addr, imm=payload:
                              ep.send_to(addr(dst), tag, payload).await (endpoint.rs:69-72,120-133)  */
    MS_OP_REPLY = 22,      /* a=src sock, b=(tag<<8), imm=payload: ep.send_to(from, tag, ..).await   */
    MS_OP_RECV = 23,       /* a=sock, b=(tag<<8): (val, from) = ep.recv_from(tag).await
                              (endpoint.rs:87-94,140-149)                                            */
    MS_OP_ASSERT_VAL = 24, /* assert_eq!(val, imm)                                                   */
    MS_OP_RECV_TIMEOUT = 25,/* a=sock, b=(tag<<8)|secs, imm=ns: timeout(secs s + ns, ep.recv_from(tag)).await;
                              on Err(Elapsed) val := MADSIM_VAL_TIMEOUT (time/mod.rs:128-140)         */
    MS_OP_CLOSE = 26,      /* a=sock: drop(ep) (BindGuard::drop, net/mod.rs:483-493)                 */
    /* -- supervisor: fault injection (runtime/mod.rs:276-303, net/mod.rs:164-222) -- */
    MS_OP_KILL = 30,       /* a=node: Handle::kill(node)     (task/mod.rs:356-371)                   */
    MS_OP_RESTART = 31,    /* a=node: Handle::restart(node)  (task/mod.rs:374-401)                   */
    MS_OP_PAUSE = 32,      /* a=node: Handle::pause(node)    (task/mod.rs:404-410)                   */
    MS_OP_RESUME = 33,     /* a=node: Handle::resume(node)   (task/mod.rs:413-424)                   */
    MS_OP_CLOG_NODE = 34,  /* a=node, b=dir(1 in,2 out,3 both): NetSim::clog_node{,_in,_out}         */
    MS_OP_UNCLOG_NODE = 35,/* a=node, b=dir                                                          */
    MS_OP_CLOG_LINK = 36,  /* a=src node, b=dst node: NetSim::clog_link (network.rs:184-189)         */
    MS_OP_UNCLOG_LINK = 37,/* a=src node, b=dst node                                                 */
    MS_OP_ASSERT_EXIT = 38,/* a=node, b=expected: assert_eq!(Handle::is_exit(node), b)               */
    MS_OP_SET_LOSS = 39,   /* a=index into madsim_config_t.loss_table: NetSim::update_config(|c|
                              c.packet_loss_rate = ..) (net/mod.rs:138-141)                          */
    MS_OP_SLEEP_RAND = 40, /* sleep(thread_rng().gen_range(a*50 ms .. b s + imm ns)).await: the randomised
                              fault loop of tonic-example/tests/test.rs:198-201 (UniformDuration, A.3) */
    /* -- shared test flags: the Arc<AtomicUsize> the reference's tests observe (task/mod.rs:864-1015) -- */
    MS_OP_GSET = 41,       /* a=flag(0..3): flag.store(imm)                                         */
    MS_OP_GADD = 42,       /* a=flag: flag.fetch_add(imm)                                            */
    MS_OP_ASSERT_G = 43,   /* a=flag: assert_eq!(flag.load(), imm)                                   */
    MS_OP_PANIC_IF_G_LT = 44, /* a=flag: if flag.load() < imm { panic!() }                            */
    MS_OP_JEQ = 45,        /* b=target: if val == imm { goto b } (react to a timeout / a reply value)    */
    /* -- reliable channel: what madsim-tonic / etcd / kafka sims ride on (net/mod.rs:337-430) -- */
    MS_OP_CONNECT = 46,    /* a=ep, b=dst addr: (tx, rx) = ep.connect1(addr).await; val := 0, or
                              MADSIM_VAL_REFUSED on ConnectionRefused (endpoint.rs:178-193)           */
    MS_OP_ACCEPT = 47,     /* a=ep: (tx, rx, _) = ep.accept1().await (endpoint.rs:197-211)           */
    MS_OP_CSEND = 48,      /* imm=payload: tx.send(payload).await; val := MADSIM_VAL_RESET when the
                              peer's receiver is gone (endpoint.rs:240-245, net/mod.rs:417-421)      */
    MS_OP_CRECV = 49,      /* val = rx.recv().await: in-order delivery, 1 ms -> 10 s backoff while the link
                              is down (net/mod.rs:385-402); MADSIM_VAL_RESET when the channel closed */
    MS_OP_CCLOSE = 50,     /* drop(tx); drop(rx)                                                     */
    /* -- typed RPC over the datagram Endpoint (net/rpc.rs:96-180) -- */
    MS_OP_RPC_CALL = 51,   /* a=ep, b=(req_tag<<8)|dst addr, imm=(timeout_ms<<8)|request code:
                              val = ep.call(dst, req).await (rpc.rs:108-131): rsp_tag = random::<u64>() (one
                              RngCore draw), send_to_raw(dst, R::ID, (rsp_tag, req)), recv_from_raw(rsp_tag),
                              assert_eq!(from, dst).  timeout_ms != 0: ep.call_timeout(dst, req, d) (rpc.rs:96-105),
                              val := MADSIM_VAL_TIMEOUT on Err(TimedOut).  req_tag must be a typed tag (>= 0x80).   */
    MS_OP_RPC_REPLY = 52,  /* a=ep, imm=response code: net.send_to_raw(from, rsp_tag, rsp) of the request this
                              task received (or inherited at spawn): the tail of add_rpc_handler (rpc.rs:170-176) */
    MS_OP_RAND_BOOL = 53,  /* a=index into madsim_config_t.loss_table: val = thread_rng().gen_bool(p) as u32 — one
                              RngCore draw unless p == 1 (madsim-etcd-client/src/service.rs:165-166)        */
    MS_OP_RANDOM = 54,     /* a=0: val = thread_rng().gen::<u32>() (rand.rs:142-158, one with());
                              a=1: val = the byte of getrandom(&mut [0u8; 1]) (rand.rs:197-211: with(|r| r.fill_bytes(buf))) */
    MS_OP_TRACE_TIME = 55, /* a=0: obs_hash <- fold(SystemTime::now() in ns since UNIX_EPOCH): the per-seed base time
                              (time/mod.rs:26-33) + elapsed; a=1: fold(Instant elapsed ns) (system_time.rs:122-154);
                              a=2: fold(val): make the last received / drawn value observable                      */
    /* -- NetSim message hooks (net/mod.rs:240-284; consulted by NetSim::send, :307-311 and :323-328) -- */
    MS_OP_HOOK_REQ = 56,   /* a=node, b=(req_tag<<8)|mode, imm=request code: NetSim::hook_rpc_req::<R>(node, f) — replaces the
                              node's request hook.  f returns false (the request is dropped after rand_delay, BEFORE the link
                              test: no loss / latency draws, no delivery timer) for typed-RPC requests of tag req_tag sent FROM
                              `node` whose code == imm (mode 0) or for all of them (mode 1); every other payload passes.    */
    MS_OP_HOOK_RSP = 57,   /* a=node, b=mode, imm=response code: NetSim::hook_rpc_rsp::<R>(node, f) — replaces the node's response
                              hook.  The hook installed when a typed-RPC response is SENT towards `node` judges it when its
                              delivery timer fires: false (code == imm in mode 0, any response in mode 1) = the timer fires and
                              nothing is delivered.  Responses carry no type id here: the hook sees every response.           */
    /* -- IP Virtual Server at run time (net/ipvs.rs:50-85): the calls a test makes on NetSim::global_ipvs() -- */
    MS_OP_IPVS = 58,       /* a=MADSIM_IPVS_*, b=service (index into madsim_workload_t.services), imm=socket-table entry of a
                              server address (ADD_SERVER / DEL_SERVER).  ADD_SERVICE = HashMap::insert of a fresh Service (no
                              servers, rr_index 0: also when the service exists); DEL_SERVICE removes it; ADD_SERVER pushes
                              (`.expect("service not found")`: the calling task panics when the service is absent);
                              DEL_SERVER = servers.retain(|a| a != server_addr), rr_index untouched — get_server resets an
                              index that ran off the end (ipvs.rs:96-98).  No draw, no await.                              */
    /* -- ABI v4 -- */
    MS_OP_SET_LATENCY = 59,/* a=index into madsim_config_t.lat_table: NetSim::update_config(|c| c.send_latency = lo..hi)
                              (net/mod.rs:138-141 -> Network::update_config, net/network.rs:129): every later link test samples
                              the new range (network.rs:267) — datagram sends, connect1, channel sends and their retries alike;
                              messages already in flight keep the latency they drew.  No draw, no await.                    */
    MS_OP__COUNT
};
#define MADSIM_IPVS_ADD_SERVICE 0u
#define MADSIM_IPVS_DEL_SERVICE 1u
#define MADSIM_IPVS_ADD_SERVER  2u
#define MADSIM_IPVS_DEL_SERVER  3u

/* Typed RPC (net/rpc.rs): request tags R::ID are the tag values 0x80..0xFD; a message received on one carries the
 * caller's response tag besides its 8-bit request code (val), and MS_OP_SPAWN with MADSIM_SPAWN_MOVE_REQUEST hands
 * (val, from, response tag) to the per-request task like `spawn(async move { .. })` in rpc.rs:170.  Response tags
 * are u64 draws in the reference; here a response is matched to the caller's pending receive itself, which is
 * what an unguessable tag means.  Request and response codes are 8 bits in this build. */
#define MADSIM_TAG_RPC_FIRST 0x80u
#define MADSIM_TAG_RPC_LAST  0xFDu
#define MADSIM_SPAWN_MOVE_CONN    2u   /* MS_OP_SPAWN b: the (tx, rx) pair moves into the child               */
#define MADSIM_SPAWN_MOVE_REQUEST 4u   /* MS_OP_SPAWN b: the received request moves into the child             */

#define MADSIM_VAL_TIMEOUT 0xFFFFFFFFu
#define MADSIM_VAL_REFUSED 0xFFFFFFFEu
#define MADSIM_VAL_RESET   0xFFFFFFFDu
#define MADSIM_VAL_ADDR_NOT_AVAILABLE 0xFFFFFFFCu
#define MADSIM_VAL_ADDR_IN_USE        0xFFFFFFFBu

/* A task program: where it runs and where it starts.  Program 0 is the body handed to block_on
 * (the main task on node 0, task/mod.rs:222-235). */
typedef struct madsim_prog {
    uint8_t  node;   /* 0 = supervisor node "madsim-main"; 1..n_nodes = create_node() order          */
    uint8_t  flags;  /* MADSIM_PROG_* */
    uint16_t entry;  /* first instruction                                                            */
} madsim_prog_t;
#define MADSIM_PROG_INIT 1u /* NodeBuilder::init task: spawned at build() (MS_OP_BUILD, or before
                               block_on when MADSIM_PROG_PRE is also set) and on every restart
                               (runtime/mod.rs:405-418, task/mod.rs:395-400,472-474)                 */
#define MADSIM_PROG_PRE  2u /* spawned before Runtime::block_on pushes the main task, in prog order:
                               `node.spawn(..)` ahead of `runtime.block_on(..)` (task/mod.rs:859-897) */
#define MADSIM_PROG_DROP_SPAWN 4u /* the body owns a guard moved into it whose Drop calls task::spawn(program p + 1)
                               (the reference tests spawn_in_future_drop_by_aborting_task / _by_killing_node,
                               task/mod.rs:1185-1253): whenever an instance finishes — returns, or its future is
                               dropped because it was aborted, its node killed or it panicked — after its other locals
                               have dropped and before its JoinHandle awaiter is notified.  The child is spawned in the
                               dying task's context (run_all_ready enters it for `drop` as for `run`, :284-287): same
                               node, same NodeInfo incarnation, so a guard dropped by a kill spawns a task that is
                               scheduled once and dropped unpolled.  Program p + 1 must run on the same node; not
                               combined with MS_OP_PAUSE (a parked Runnable is dropped in the KILLER's context).    */

/* A socket address an Endpoint may bind or a datagram may be sent to: an IP and a port.  `kind` picks the IP: the node's
 * own 10.0.0.<node>, 0.0.0.0 or 127.0.0.1 as used ON `node` (entries of the last two kinds are per node: only tasks of
 * `node` may bind them).  Resolution happens at try_send time exactly as
 * in network.rs:272-313: loopback or an exact match among the sender's own sockets keeps the message on the sender's
 * node; an IP-less sender or an unknown IP drops it WITHOUT any RNG draw; after the link test (loss + latency draws,
 * msg_count) the destination node's sockets are searched for the exact address, then for 0.0.0.0:port.  The address a
 * receiver sees (`from`, what MS_OP_REPLY answers to) is the sender's real IP — or 127.0.0.1 when the datagram was sent
 * to a loopback address — with the sending socket's port (network.rs:307-311).
 * port == 0 is an EPHEMERAL Endpoint (`Endpoint::bind("0.0.0.0:0")`): every MS_OP_BIND of the entry takes the lowest port
 * from 1 up that no socket of the node holds for that IP (network.rs:224-236) and every other op on the entry works on the
 * Endpoint its last bind made.  Such an entry is not an address anybody can name: it cannot be a destination operand
 * (MS_OP_SEND / MS_OP_CONNECT / MS_OP_RPC_CALL `b`) — peers reach it by replying to `from`, or through a named entry that
 * happens to carry the port it was given.
 * The device table holds one candidate entry per port such an Endpoint can get — as many as the node has entries for that
 * IP — and the total, candidates included, is limited to 63 entries; a workload that keeps more Endpoints of one entry
 * alive than that gets the resource-overflow verdict. */
typedef struct madsim_sock {
    uint8_t  node;
    uint8_t  kind;   /* MADSIM_ADDR_* */
    uint16_t port;
} madsim_sock_t;
#define MADSIM_ADDR_IP          0u  /* 10.0.0.<node>:port */
#define MADSIM_ADDR_UNSPECIFIED 1u  /* 0.0.0.0:port       */
#define MADSIM_ADDR_LOOPBACK    2u  /* 127.0.0.1:port     */
#define MADSIM_ADDR_VIRTUAL     3u  /* a virtual IP that belongs to no node ("1.1.1.1:80"): `node` is an id of the IP (1..255), not a
                                       node.  Only a destination: a workload that binds it is refused (MADSIM_E_WORKLOAD), and a
                                       datagram sent to it is dropped without any draw ("destination not found", :285-289) unless an
                                       IPVS service rewrites the destination first (madsim_service_t)                                */

/* IP Virtual Server (net/ipvs.rs): `NetSim::global_ipvs().add_service(ServiceAddr::Tcp(vaddr), RoundRobin)` plus one
 * `add_server` per entry of `servers`, done before the first task runs (as the reference's own test does,
 * net/tcp/mod.rs:254-315).  NetSim::send and connect1 ask `ipvs.get_server(dst)` after rand_delay and the request hook and
 * before Network::try_send (net/mod.rs:312-317, :345-350): when `dst` equals a service's virtual address and the service has
 * servers, the destination becomes servers[rr_index] and the per-seed rr_index advances (ipvs.rs:88-105) — also when the
 * message is then lost, clogged or finds no socket.  Everything downstream sees the rewritten address: the link test, the
 * socket lookup, the `dst` a connection's channel() halves are built from.  (An RPC caller still asserts `from == dst` with
 * the address it was GIVEN, rpc.rs:126: a typed call through a virtual address panics when the real server answers, as in
 * the reference.)  The table gives every service's address and its state before the first task runs; MS_OP_IPVS changes
 * that state at run time (add_service / del_service / add_server / del_server, ipvs.rs:50-85), per seed, at most 6 servers
 * per service at any time (a seventh add_server yields MADSIM_UNSUPPORTED: no limit grows that word). */
typedef struct madsim_service {
    uint8_t vaddr;       /* socket-table entry holding the service address (any kind; typically MADSIM_ADDR_VIRTUAL) */
    uint8_t n_servers;   /* 0..6 real servers, in add_server order; 0 = get_server() returns None: no rewrite.
                            | MADSIM_SERVICE_ABSENT: only the address is declared — the service does not exist until a task
                            runs MS_OP_IPVS ADD_SERVICE on it (n_servers & 7 must be 0 then)                          */
    uint8_t servers[6];  /* socket-table entries of the real server addresses                                       */
} madsim_service_t;
#define MADSIM_MAX_SERVICES 8u
#define MADSIM_SERVICE_ABSENT 0x80u

typedef struct madsim_node {
    uint8_t flags;       /* MADSIM_NODE_* */
    uint8_t n_match;     /* MADSIM_NODE_RESTART_MATCHING: number of patterns (0..2)                  */
    uint8_t match[2];    /* the panic message codes that restart the node                            */
} madsim_node_t;
#define MADSIM_NODE_RESTART_ON_PANIC 1u /* NodeBuilder::restart_on_panic (task/mod.rs:298-316)      */
#define MADSIM_NODE_RESTART_MATCHING 4u /* NodeBuilder::restart_on_panic_matching(msg) (runtime/mod.rs:384-387,
                                           task/mod.rs:299): restart when the panic's message code is one of
                                           `match[0..n_match)` — or, with madsim_workload_t.panic_match, when the node's
                                           256-bit row has the code's bit set (the host evaluates `contains(pattern)` per code:
                                           substring patterns against literal AND run-time formatted messages)            */
#define MADSIM_PANIC_CODE_OTHER 255u
#define MADSIM_NODE_NO_IP 2u /* create_node() without .ip(..): binds any address, cannot send to another node's IP
                                (network.rs:281-283: "ip not set", no RNG draw)                              */

typedef struct madsim_workload {
    uint32_t n_nodes;   /* nodes 1..n_nodes besides node 0                                           */
    uint32_t n_progs;   /* prog 0 = main                                                             */
    uint32_t n_socks;
    uint32_t n_insns;
    const madsim_node_t* nodes;  /* [n_nodes+1], index = node id                                     */
    const madsim_prog_t* progs;  /* [n_progs]                                                        */
    const madsim_sock_t* socks;  /* [n_socks]                                                        */
    const madsim_insn_t* insns;  /* [n_insns]                                                        */
    /* -- ABI v3 -- */
    uint32_t n_services;         /* IPVS virtual services (<= MADSIM_MAX_SERVICES); their use selects the general-address build */
    uint32_t panic_dyn_max;      /* largest message code a run-time formatted panic (MS_OP_PANIC a=1) may produce; 0 = 254.
                                    A workload that also uses literal messages gives those the codes above it; a formatted value
                                    beyond it yields MADSIM_UNSUPPORTED for the seed (never a silently different answer)       */
    const madsim_service_t* services;   /* [n_services] or NULL                                                       */
    const uint32_t* panic_match; /* NULL, or [n_nodes+1][8]: bit c of row n = "a panic whose message code is c restarts node n"
                                    (NodeBuilder::restart_on_panic_matching, `error_msg.contains(pattern)` task/mod.rs:297-300,
                                    evaluated by the host for every code: literal messages are interned as codes, a formatted
                                    message is the decimal text of its code).  Rows of nodes without MADSIM_NODE_RESTART_MATCHING
                                    are ignored.  NULL: the rows are built from madsim_node_t.match[] (equality on the code).   */
} madsim_workload_t;

/* ------------------------------------------------------------------------------------------------
 * Inputs mirroring Builder's public fields (runtime/builder.rs:7-22) and net Config
 * (net/network.rs:66-89).
 * ---------------------------------------------------------------------------------------------- */
typedef struct madsim_config {
    double   packet_loss_rate;   /* Config.net.packet_loss_rate, default 0.0                         */
    uint64_t lat_lo_ns;          /* Config.net.send_latency.start, default 1 ms                      */
    uint64_t lat_hi_ns;          /* Config.net.send_latency.end (exclusive), default 10 ms           */
    uint32_t buggify;            /* non-zero: buggify enabled for the whole run (rand.rs:113-134)    */
    uint32_t n_loss_table;       /* entries in loss_table used by MS_OP_SET_LOSS                     */
    double   loss_table[4];
    /* -- ABI v4 -- */
    uint32_t n_lat_table;        /* entries in lat_table_* used by MS_OP_SET_LATENCY (<= 4); every entry must be a non-empty range
                                    (`gen_range` of an empty range panics in the reference, network.rs:267), and an op that names
                                    an entry >= n_lat_table is refused (MADSIM_E_WORKLOAD)                                   */
    uint32_t reserved0;          /* 0 */
    uint64_t lat_table_lo_ns[4]; /* send_latency.start of entry i                                    */
    uint64_t lat_table_hi_ns[4]; /* send_latency.end (exclusive) of entry i                          */
} madsim_config_t;

#define MADSIM_LIMIT_NONE 0xffffffffu  /* a capacity of zero (0 itself means "pick a default") */

typedef struct madsim_limits {
    uint64_t time_limit_ns;      /* Builder.time_limit; 0 = None (task/mod.rs:253-258)               */
    uint32_t max_steps;          /* device safety net; 0 = default (1<<24). Not a reference concept. */
    uint32_t heap_lds_slots;     /* timer-heap entries kept in LDS per seed; 0 = auto                */
    uint32_t heap_spill_slots;   /* further entries per seed in the coalesced HBM spill region       */
    uint32_t max_tasks;          /* live task instances per seed; 0 = auto (n_progs + restarts)      */
    uint32_t mbox_regs;          /* pending recv registrations per socket; 0 = auto (2)              */
    uint32_t mbox_msgs;          /* undelivered messages per socket; 0 = auto (2), MADSIM_LIMIT_NONE = none */
    uint32_t lanes_per_wave;     /* seeds carried per 64-lane wave (8/16/32/64); 0 = auto.  With state_mem = MADSIM_STATE_GLOBAL: 64,
                                    or 32 for timeout-only workloads (twice the LDS per seed: more of the timer heap out of the
                                    spill region — the election loop's setting); anything else there is MADSIM_E_LIMITS */
    uint32_t max_conns;          /* live reliable-channel connections per seed; 0 = auto (4)         */
    uint32_t chan_queue;         /* queued payloads per channel direction; 0 = auto (2)              */
    uint32_t sched;              /* MADSIM_SCHED_*: how lanes pick up further seeds when a launch holds more
                                    seeds than resident lanes; 0 = static striding                    */
    uint32_t state_mem;          /* MADSIM_STATE_*: where a seed's task table and planes live; 0 = auto               */
    uint32_t max_steps_ceiling;  /* the largest step cap a re-run of MADSIM_STEP_LIMIT seeds (madsim_hip_run_batch_auto / _multi)
                                    may use; 0 = default (1 << 28: the first pass's 1 << 24, then ONE 16x round).  A seed that
                                    still hits the cap keeps the MADSIM_STEP_LIMIT verdict: a livelocked workload (a yield or
                                    1 ms timer loop without a time limit) costs seconds, not one hour-long kernel              */
    uint32_t no_trace_hash;      /* non-zero: do not fold the determinism log into madsim_result_t.trace_hash (it comes back 0).
                                    The reference computes a log byte per RNG call only while check_determinism logs or checks
                                    (rand.rs:67 `if lock.log.is_some() || lock.check.is_some()`); the per-seed fingerprint of
                                    that log in every result is an extra of this runner — 0 keeps it (default), a plain
                                    Builder::run needs none of it (measured: 4 % faster on the ping-pong kernel; madsim_hip_trace_seed always logs) */
} madsim_limits_t;

#define MADSIM_STATE_AUTO   0u   /* LDS unless an extended-op workload's state leaves a CU fewer than 4 full waves       */
#define MADSIM_STATE_LDS    1u   /* all per-seed state in LDS ([word][lane] planes)                                       */
#define MADSIM_STATE_GLOBAL 2u   /* extended-op workloads: task table + planes in global memory ([unit][lane] across the launch:
                                    L2 / Infinity Cache / HBM), only the timer-heap top and the ready queue in LDS.  A workload
                                    of base ops only has no such build and keeps its state in LDS (the setting is ignored)  */

#define MADSIM_STATE_COMPACT 3u  /* base-op workloads on full waves, <= 8 tasks, no heap spill, sleeps < 2.1 s: 8-byte timer-heap entries (low
                                    deadline word: exact inside that horizon), heap root in registers, the main task in global memory —
                                    a fourth wave per SIMD for the 4-node ping-pong.  AUTO takes it when it gains a workgroup per CU  */

/* OR-ed into madsim_limits_t.state_mem (the low byte keeps MADSIM_STATE_*): global-state builds of timeout-only workloads keep the
   re-registrations of a pending Sleep — time/sleep.rs:51-53 registers ANOTHER timer with the same deadline and waker on every
   not-elapsed poll — as a COUNT beside the first entry instead of as further heap entries (k_timer.h dedup_note).  Identical
   entries fire back to back and every one after the first is a no-op wake-up, so the result is the same bit for bit unless two
   DIFFERENT events tie on a deadline (their order depends on the heap's shape): the kernel notices such a tie when it pops it and
   runs that seed again from the start with the literal heap, so callers never see a difference — only time.  Ignored by the other
   builds.  Worth it where timeouts dominate and ties are rare (the election loop: 34 % of its timer pushes are such duplicates,
   5e-4 of its seeds meet a tie). */
#define MADSIM_STATE_DEDUP_TIMERS 0x100u

/* OR-ed into madsim_limits_t.state_mem like MADSIM_STATE_DEDUP_TIMERS: global-state builds with a heap-spill region keep every timer-heap
   entry as 8 bytes {low 32 bits of the deadline, event word} instead of 16 — twice the heap levels in the same LDS, half the bytes per
   spilled level; the payload of a datagram delivery waits in a per-seed record pool in global memory until its entry reaches the root.
   Exact while every live deadline lies within 2^31 ns (2.1 s) of the clock: the host admits the layout only for workloads whose sleeps,
   timeouts and latencies stay below that (no buggify, no restart_on_panic), and the device checks every push — a seed that does exceed
   it (a channel back-off past 2 s) gets a capacity verdict and is run again on the 16-byte entries by madsim_hip_run_batch_auto.  Results
   never differ.  Ignored by builds without the variant (LDS-resident state, connection-only workloads, general address resolution). */
#define MADSIM_STATE_NARROW_HEAP 0x200u

#define MADSIM_SCHED_STATIC 0u   /* lane g runs seeds g, g+G, g+2G, ...                                */
#define MADSIM_SCHED_QUEUE  1u   /* a finished lane pulls the next seed from a per-launch atomic counter */

/* ------------------------------------------------------------------------------------------------
 * Outputs.
 * ---------------------------------------------------------------------------------------------- */
enum madsim_verdict {
    MADSIM_PASS = 0,        /* block_on returned                                                     */
    MADSIM_PANIC = 1,       /* a task panicked and its node does not restart (task/mod.rs:315)       */
    MADSIM_DEADLOCK = 2,    /* "no events, all tasks will block forever" (task/mod.rs:250)           */
    MADSIM_TIME_LIMIT = 3,  /* "time limit exceeded" (task/mod.rs:253-258)                           */
    MADSIM_OVERFLOW = 4,    /* a device capacity in madsim_limits_t was exceeded: re-run the seed with
                               larger limits (not a reference verdict; never silently wrong).  Capacities a larger limit
                               CAN lift give this verdict (at their ceilings the same events are MADSIM_UNSUPPORTED, below) — with
                               two exceptions that stay capacity verdicts at the ceiling, where a re-run cannot help: the timer
                               heap at 2^20 spilled entries, and the 128th message queued in one mailbox of a workload without
                               extended ops (its header keeps 7 bits).  The FIRST capacity or model event of a seed decides
                               its runner verdict (what follows it in the same round runs on spoiled state)            */
    MADSIM_STEP_LIMIT = 5,  /* max_steps reached (not a reference verdict)                           */
    MADSIM_UNSUPPORTED = 6, /* the seed left the workload MODEL (not a reference verdict, and larger limits do not help:
                               never re-run): a port-0 table entry bound again while the Endpoint of its previous bind is
                               alive — an entry names one Endpoint at a time, `close` it first; a seventh server added to
                               an IPVS service; a ninth connection waiting in one Endpoint's accept1 queue; a 255th live task,
                               a 256th registration of one socket, a 16th queued channel payload, a 128th live connection, a 256th
                               message queued in one mailbox (the ceilings of max_tasks / mbox_regs / chan_queue / max_conns /
                               mbox_msgs: below them the verdict is MADSIM_OVERFLOW and a re-run lifts it); a 128th connection end
                               holding one Endpoint's address; an ephemeral bind when every candidate port the table has for that
                               (node, IP) is taken (addresses kept alive by connections of Endpoints long dropped); a new
                               receive whose registration word — 8 bits of the task's receive count, 8 of its slot's generation —
                               equals a dead registration's still in the list (256 receives of one task while a timed-out one
                               lingers, 256 instances of one slot); a formatted panic value above
                               madsim_workload_t.panic_dyn_max.  The oracle reports the same verdict for the same seed, decided at
                               the same instruction; every other result field is 0                                          */
    MADSIM_INTERNAL = 7     /* an invariant of the device code broke (a bug in this library, never a property of the
                               workload): the parity tests assert that no seed ever carries it; other fields 0        */
};
#define MADSIM_MAX_LIVE_TASKS 254u   /* the ceiling of madsim_limits_t.max_tasks: a 255th live task is MADSIM_UNSUPPORTED (kernel and oracle) */
#define MADSIM_MAX_MBOX_REGS 255u    /* the ceiling of mbox_regs: a 256th pending / dead registration of one socket is MADSIM_UNSUPPORTED          */
#define MADSIM_MAX_CHAN_QUEUE 15u    /* the ceiling of chan_queue: a 16th payload queued in one channel direction is MADSIM_UNSUPPORTED             */
#define MADSIM_MAX_CONNS 127u        /* the ceiling of max_conns: a 128th live connection is MADSIM_UNSUPPORTED                                     */
#define MADSIM_MAX_MBOX_MSGS 255u    /* the ceiling of mbox_msgs (workloads with extended ops; 127 without): a 256th message queued in one mailbox is MADSIM_UNSUPPORTED */
#define MADSIM_MAX_SOCKET_GUARDS 127u /* connection ends (Sender + Receiver pairs) that may hold one Endpoint's BindGuard at a time: the 128th is MADSIM_UNSUPPORTED */
/* verdicts >= MADSIM_OVERFLOW are RUNNER verdicts: statements about this runner, never a test's failure */
#define MADSIM_IS_RUNNER_VERDICT(v) ((v) >= MADSIM_OVERFLOW)

/* 48 bytes per seed.  Everything here is compared bit-for-bit against the oracle. */
typedef struct madsim_result {
    uint32_t verdict;
    uint32_t steps;       /* executor steps: task polls (incl. dropped runnables) + timer fires      */
    uint64_t clock_ns;    /* Clock.elapsed() when block_on left (time/mod.rs:230-233)                */
    uint64_t msg_count;   /* NetSim::stat().msg_count (network.rs:99-105,265)                        */
    uint64_t rng_calls;   /* Xoshiro256PlusPlus::next_u64 invocations on the GlobalRng               */
    uint64_t trace_hash;  /* FNV-1a-64 over the determinism-log bytes of rand.rs:64-88               */
    uint64_t obs_hash;    /* FNV-1a-64 over MS_OP_TRACE values in execution order                    */
} madsim_result_t;

typedef struct madsim_summary {
    uint64_t first_failing_seed; /* minimum seed with verdict != PASS; UINT64_MAX if none — test n_failed, a
                                    batch may legitimately contain seed UINT64_MAX itself            */
    uint64_t n_failed;
    uint64_t total_steps;
    uint64_t total_clock_ns;
    double   kernel_ms;          /* HIP-event time of the simulation kernel(s) on the call's stream   */
    double   wall_s;             /* host wall time of the call                                       */
} madsim_summary_t;

/* ------------------------------------------------------------------------------------------------
 * Entry points.  All return 0 on success, <0 on error (madsim_hip_strerror); they never unwind.
 * Per-seed failures are data (verdict), not errors.
 * ---------------------------------------------------------------------------------------------- */
#define MADSIM_E_ARG      (-1)
#define MADSIM_E_HIP      (-2)
#define MADSIM_E_NOINIT   (-3)
#define MADSIM_E_WORKLOAD (-4)
#define MADSIM_E_LIMITS   (-5)

uint32_t    madsim_hip_version(void);
/* Identity of the loaded library: "madsim_hip abi=<MADSIM_HIP_ABI_VERSION> arch=gfx950 kernels=<builds> ..." — the product is
 * the gfx950 HIP build and nothing else; a host mirror that lets the library path be overridden (A/B builds) checks this
 * string so that no other object exporting the same symbols can stand in for it. */
const char* madsim_hip_build_info(void);
const char* madsim_hip_strerror(int code);
const char* madsim_hip_last_error(void);   /* thread-local text of the calling thread's last failure */
/* The library keeps up to five batches of one call in flight on its own HIP streams; ROCclr maps a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run one after the other (madsim_hip_run_batch(262 144):
 * 8.3 ms against ~5 ms).  The library never touches the environment by itself: set GPU_MAX_HW_QUEUES=16 before the process first uses
 * HIP, or call this FIRST THING in main() (it is a setenv: not safe beside threads that read the environment, no effect once HIP is
 * initialised).  Returns 1 when it set the variable, 0 when the host's own setting stands, <0 on a bad argument. */
int madsim_hip_prefer_hw_queues(int n);

/* ---- Per-device contexts -----------------------------------------------------------------------------
 * "one host thread drives a batch; library is re-entrant per device handle, not thread-safe per
 * handle".  A context owns everything the runner keeps on one GPU (workload tables, spill regions, events);
 * distinct contexts share no mutable state, so a process may hold one per GPU and use them from different threads,
 * or drive all of them from one thread with madsim_hip_run_batch_multi.  Calls on ONE context are serialised.
 * Every madsim_hip_ctx_* function mirrors the v1 function of the same name below, which runs on the
 * process-default context created by madsim_hip_init. */
typedef struct madsim_hip_ctx madsim_hip_ctx_t;
int madsim_hip_ctx_create(int device, madsim_hip_ctx_t** out);
int madsim_hip_ctx_destroy(madsim_hip_ctx_t* ctx);
int madsim_hip_ctx_device(const madsim_hip_ctx_t* ctx);
madsim_hip_ctx_t* madsim_hip_default_ctx(void);   /* NULL before madsim_hip_init */

/* Bind the calling process to one GPU (one process per GPU; replaces nothing in the reference —
 * the reference's per-seed std::thread::spawn, builder.rs:134, has no device). */
int madsim_hip_init(int device);
int madsim_hip_shutdown(void);

/* Replaces the seed loop of Builder::run (builder.rs:129-160) for `count` seeds from `seed0`.
 * `out` is caller-allocated host memory [count] (may be NULL when only the summary is wanted). */
int madsim_hip_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg,
                         uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                         madsim_result_t* out, madsim_summary_t* summary);

/* madsim_hip_run_batch, then every seed that came back with a RUNNER verdict — MADSIM_OVERFLOW (a device capacity)
 * or MADSIM_STEP_LIMIT (the max_steps safety net) — is run again, all of them gathered into ONE compacted launch per
 * round, with doubled capacities / a 16x step cap (never above madsim_limits_t.max_steps_ceiling), up to `max_rounds` times; the summary is
 * recomputed over the final results.  The reference's containers are unbounded (Vec mailboxes, BinaryHeap timers) and
 * it has no step cap, so neither is ever a test verdict: this is the entry point a Builder::run replacement calls.
 * `out` must not be NULL. */
int madsim_hip_run_batch_auto(const madsim_workload_t* w, const madsim_config_t* cfg,
                              uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                              madsim_result_t* out, madsim_summary_t* summary, int max_rounds);

/* Same, results stay resident in HBM: `d_out` is device memory [count] owned by the caller,
 * `stream` is a hipStream_t (NULL = the null stream).  Asynchronous unless `summary` != NULL
 * (the summary needs a device reduction + D2H of 32 bytes, which synchronises `stream`). */
int madsim_hip_run_batch_device(const madsim_workload_t* w, const madsim_config_t* cfg,
                                uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                                void* d_out, void* stream, madsim_summary_t* summary);

/* Fully asynchronous form: nothing is copied to the host.  `d_summary4` (device memory, 4 x uint64_t, may be NULL)
 * receives {first failing seed ^ (1 << 63), n_failed, total_steps, total_clock_ns} from a reduction kernel queued behind
 * the simulation on `stream`.  Word 0 is in order-preserving int64 form (none = INT64_MAX), so a signed RCCL
 * all-reduce(MIN) on it and all-reduce(SUM) on words 1-3 combine ranks without touching the host.  `timing_slot` in [0,64) brackets
 * the simulation kernel with a HIP event pair readable through madsim_hip_timing_ms after the stream has run; -1 = none. */
int madsim_hip_run_batch_async(const madsim_workload_t* w, const madsim_config_t* cfg,
                               uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                               void* d_out, void* d_summary4, void* stream, int timing_slot);
int madsim_hip_timing_ms(int timing_slot, double* ms);

/* Re-run one seed and return the raw determinism log (rand.rs:64-88 byte per GlobalRng::with),
 * the artefact MADSIM_TEST_CHECK_DETERMINISM compares (runtime/mod.rs:178-202).  Returns the number
 * of bytes the log holds (may exceed cap; only cap bytes are written), <0 on error. */
int64_t madsim_hip_trace_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                              const madsim_limits_t* lim, uint8_t* log, uint64_t cap,
                              madsim_result_t* out);

/* The same entry points on an explicit context (see "Per-device contexts" above). */
int madsim_hip_ctx_run_batch(madsim_hip_ctx_t* ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                             uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                             madsim_result_t* out, madsim_summary_t* summary);
int madsim_hip_ctx_run_batch_auto(madsim_hip_ctx_t* ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                                  uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                                  madsim_result_t* out, madsim_summary_t* summary, int max_rounds);
int madsim_hip_ctx_run_batch_device(madsim_hip_ctx_t* ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                                    uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                                    void* d_out, void* stream, madsim_summary_t* summary);
int madsim_hip_ctx_run_batch_async(madsim_hip_ctx_t* ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                                   uint64_t seed0, uint64_t count, const madsim_limits_t* lim,
                                   void* d_out, void* d_summary4, void* stream, int timing_slot);
int madsim_hip_ctx_timing_ms(madsim_hip_ctx_t* ctx, int timing_slot, double* ms);
int64_t madsim_hip_ctx_trace_seed(madsim_hip_ctx_t* ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                                  uint64_t seed, const madsim_limits_t* lim, uint8_t* log, uint64_t cap,
                                  madsim_result_t* out);

/* One process, several GPUs: the whole seed loop of Builder::run (builder.rs:129-150, every seed driven from one
 * process) over `n_ctx` contexts, each on its own GPU (or, for tests on a 1-GPU box, several on the same one).
 * Seeds are sharded contiguously — context g runs [seed0 + g*ceil(count/n_ctx), ...) —, every device's kernel is
 * queued from the calling thread before any is waited for, results land in `out[count]` (host), seeds that outgrew a
 * device capacity or the step cap are re-run in one compacted launch per round (<= max_rounds, as
 * madsim_hip_run_batch_auto), and the n reports are folded on the host into `summary`.  Bit-identical to
 * madsim_hip_run_batch_auto on a single context.  summary.kernel_ms = the MAXIMUM over the devices (they run concurrently) plus the
 * re-run rounds; madsim_campaign_t.kernel_ms, also over several contexts, is the SUM of every batch's kernel time — the two are not
 * comparable. */
int madsim_hip_run_batch_multi(madsim_hip_ctx_t* const* ctxs, int n_ctx, const madsim_workload_t* w,
                               const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                               const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary,
                               int max_rounds);

/* ---- Campaigns: many batches, kept in flight by the library -------------------------------------------------------------
 * One 65 536-seed batch is one wave per SIMD: alone it leaves two thirds of the issue slots idle (a base-op launch takes ~3.2 ms
 * alone, ~1.3 ms per batch with three in flight).  madsim_hip_run_batch cannot overlap anything — it returns the results —, so
 * the overlap lives here: `total` seeds from `seed0` run as batches of `batch` seeds (0 = 65 536) on the context's own HIP
 * streams, `in_flight` at a time (0 = auto: 3; 5 when the workload's LDS admits four waves per SIMD; 4 for a global-state build with a heap-spill region; more streams than hardware queues are pointless: GPU_MAX_HW_QUEUES), each followed by a
 * device reduction whose 48-byte report is the only thing copied to the host.  Per-seed results are NOT returned: a campaign
 * answers "which is the first failing seed, how many fail" — the seed-search use of `MADSIM_TEST_NUM` — and the caller re-runs
 * the seed it is told about (madsim_hip_run_batch / madsim_hip_trace_seed) for details.
 * MADSIM_CAMPAIGN_STOP_AT_FAILURE: stop launching as soon as a completed batch reports a seed with a GENUINE verdict (panic /
 * deadlock / time limit); batches are contiguous and reports are read in order, so `first_failing_seed` is then the smallest
 * failing seed of [seed0, seed0 + seeds_run): at most `in_flight - 1` batches beyond the failing one have been started.
 * Runner verdicts (MADSIM_OVERFLOW / MADSIM_STEP_LIMIT) never stop a campaign and are counted apart (`n_runner`): re-run those
 * ranges with madsim_hip_run_batch_auto. */
#define MADSIM_CAMPAIGN_STOP_AT_FAILURE 1u
typedef struct madsim_campaign {
    uint64_t seeds_run;           /* seeds whose batch ran to completion and was read (a prefix of the range)                 */
    uint64_t batches_run;
    uint64_t batches_launched;    /* >= batches_run: with STOP_AT_FAILURE the batches in flight when the failure was read      */
    uint64_t first_failing_seed;  /* smallest seed with a genuine verdict among seeds_run; UINT64_MAX if none                  */
    uint64_t n_failed;            /* genuine failures among seeds_run                                                          */
    uint64_t n_runner;            /* seeds that came back with a runner verdict (not failures; to be re-run)                   */
    uint64_t total_steps, total_clock_ns;
    double   kernel_ms;           /* sum of the simulation kernels' HIP-event durations (they overlap: > wall time)            */
    double   wall_s;
} madsim_campaign_t;
int madsim_hip_ctx_run_campaign(madsim_hip_ctx_t* ctx, const madsim_workload_t* w, const madsim_config_t* cfg,
                                uint64_t seed0, uint64_t total, uint64_t batch, uint32_t in_flight, uint32_t flags,
                                const madsim_limits_t* lim, madsim_campaign_t* out);
int madsim_hip_run_campaign(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t total,
                            uint64_t batch, uint32_t in_flight, uint32_t flags, const madsim_limits_t* lim,
                            madsim_campaign_t* out);
/* The campaign over several devices from ONE host thread (one context per GPU; the multi-GPU form of `first-fail seeds per hour`):
 * batch k of the range runs on context k % n_ctx, `in_flight` batches per context (0 = auto), and the reports are read in batch order.
 * The devices advance through the seed range together, so MADSIM_CAMPAIGN_STOP_AT_FAILURE stops every device within one round of
 * batches of the first genuine failure, and the report — first_failing_seed, n_failed, n_runner, steps, clock of the prefix
 * [seed0, seed0 + seeds_run) — is the one a single context gives for the same prefix (runtime/builder.rs:129-160: the seeds are
 * seed0 .. seed0 + count whatever runs them, the first failure in seed order is the one reported).  No collective: a report is
 * 48 bytes per batch and the calling thread reads them anyway; ranks that are separate processes gather theirs over RCCL
 * (madsim_amd/dist.py).  Locks the contexts in address order, drains every stream before an error return. */
int madsim_hip_run_campaign_multi(madsim_hip_ctx_t* const* ctxs, int n_ctx, const madsim_workload_t* w,
                                  const madsim_config_t* cfg, uint64_t seed0, uint64_t total, uint64_t batch,
                                  uint32_t in_flight, uint32_t flags, const madsim_limits_t* lim, madsim_campaign_t* out);

/* Geometry the library picked for a workload (for DESIGN/bench reporting). */
typedef struct madsim_geometry {
    uint32_t lds_bytes_per_seed;
    uint32_t lds_bytes_per_block;
    uint32_t block_threads;
    uint32_t blocks_per_cu;
    uint32_t grid_blocks;
    uint32_t heap_lds_slots;
    uint32_t heap_spill_slots;
    uint32_t max_tasks;
    uint32_t lanes_per_wave;
    uint32_t variant;              /* kernel specialisation: bit0 heap spill, bit1 extended ops, bit2 ready queue in a
                                    * register, bit3 runtime lane stride, bit4 global-state build; bits 8-12 = classes of extended ops compiled in
                                    * (1 timeouts, 2 channel, 4 RPC, 8 node lifecycle, 16 general address resolution), bit 13 = built without the determinism-log
                                    * fold (madsim_limits_t.no_trace_hash on a base-op workload), bit 14 = the compact base-op layout (MADSIM_STATE_COMPACT), bit 15 = 8-byte
                                    * timer-heap entries (MADSIM_STATE_NARROW_HEAP); bits 16-19 = compile-time log2 lane
                                    * stride (15 = runtime) */
    uint32_t global_bytes_per_seed; /* size of a lane's state block in global memory (global-state builds), else 0 */
} madsim_geometry_t;
int madsim_hip_geometry(const madsim_workload_t* w, const madsim_limits_t* lim, madsim_geometry_t* g);

/* Debug aid: per-phase cycle accumulators of a profiling build of the kernel (all zero otherwise). */
int madsim_hip_debug_counters(uint64_t* out16);

/* Built-in workload of SURVEY.md §8d: N-node ping-pong, R rounds per pair.  Fills caller storage;
 * returns the number of instructions written or <0 if `cap_insns` is too small. */
int madsim_workload_pingpong(uint32_t n_nodes, uint32_t rounds, madsim_node_t* nodes /*[n_nodes+1]*/,
                             madsim_prog_t* progs /*[n_nodes+1]*/, madsim_sock_t* socks /*[n_nodes]*/,
                             madsim_insn_t* insns, uint32_t cap_insns, madsim_workload_t* w);

#ifdef __cplusplus
}
#endif
#endif /* MADSIM_HIP_H */
