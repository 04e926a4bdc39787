/*
 * madsim_oracle.c — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A single-threaded plain-C restatement of madsim's deterministic executor for workloads expressed
 * in the actor-program form of include/madsim_hip.h.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path (madsim_amd/csrc) never
 * links, imports or falls back to it.
 *
 * PARITY STATUS: "parity unpinned" for exact RNG values, heap tie order and wake order.  The
 * reference (madsim 0.2.34) cannot be built here (no rustc/cargo, no vendored crates) and its own
 * tests hold no golden vectors for this path — only relational properties, which
 * tests/test_oracle_*.py restate.  The arithmetic that lives in unvendored crates is restated from
 * the published algorithms and marked [DEP] below:
 *   rand_xoshiro 0.6 (Xoshiro256PlusPlus, SplitMix64 seeding), rand 0.8 (UniformInt::sample_single,
 *   UniformInt::sample, UniformDuration, Bernoulli), naive-timer 0.2 + alloc BinaryHeap, async-task 4.4
 *   (wake/schedule state machine), tokio 1 (oneshot wake, yield_now outside a runtime).
 * Known-answer anchors: the public xoshiro256++ / SplitMix64 vectors (tests/golden/).
 *
 * Every function cites the reference file:line (relative to /root/reference/madsim/src/sim/) it
 * follows.  Data structures are deliberately literal (Vec + swap_remove ready queue, array
 * BinaryHeap with Rust's sift order, Vec mailboxes) and deliberately different from the HIP
 * kernel's LDS layout, so agreement between the two is evidence and not a tautology.
 */
#define _POSIX_C_SOURCE 200809L
#include "madsim_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------
 * small growable vector helper
 * ---------------------------------------------------------------------------------------------- */
#define VEC(T) struct { T* p; size_t n, cap; }
#define vec_push(v, x) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? (v).cap * 2 : 8; \
    (v).p = realloc((v).p, (v).cap * sizeof *(v).p); } (v).p[(v).n++] = (x); } while (0)
#define vec_free(v) do { free((v).p); (v).p = NULL; (v).n = (v).cap = 0; } while (0)

#define FNV_OFFSET 14695981039346656037ull
#define FNV_PRIME  1099511628211ull
#define NS_PER_S   1000000000ull
#define NS_PER_MS  1000000ull

/* ------------------------------------------------------------------------------------------------
 * GlobalRng                                                           rand.rs:27-61, [DEP] A.1
 * ---------------------------------------------------------------------------------------------- */
/* Optional sink for the values a workload makes observable (MS_OP_TRACE / MS_OP_TRACE_TIME), in execution order:
 * what obs_hash folds.  Used by tools/ref_twin/compare.py to diff lists against real madsim's output. */
static __thread uint64_t* g_obs_buf; static __thread uint64_t g_obs_cap, g_obs_len;
static void obs_record(uint64_t v) { if (g_obs_buf && g_obs_len < g_obs_cap) g_obs_buf[g_obs_len] = v; g_obs_len++; }

typedef struct { uint64_t s[4]; } xoshiro_t;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* [DEP rand_core 0.6 SeedableRng::seed_from_u64 as overridden by rand_xoshiro: SplitMix64 fill] */
void oracle_seed_from_u64(uint64_t seed, uint64_t s[4]) {
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) {
        x += 0x9e3779b97f4a7c15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        s[i] = z ^ (z >> 31);
    }
}

/* [DEP rand_xoshiro 0.6 Xoshiro256PlusPlus::next_u64] */
uint64_t oracle_xoshiro_next(uint64_t s[4]) {
    uint64_t r = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * simulation state
 * ---------------------------------------------------------------------------------------------- */
enum { EV_WAKE = 1, EV_DELIVER = 2, EV_RESTART = 3 };

typedef struct {           /* naive_timer::Event: deadline + boxed callback               [DEP A.5] */
    uint64_t deadline;
    uint8_t  kind;
    uint16_t slot, gen;    /* EV_WAKE: task instance + generation (a cloned Waker)                   */
    uint8_t  sock, sockgen;            /* EV_DELIVER: captured Arc<dyn Socket>                        */
    uint32_t from;                     /* EV_DELIVER: src addr (mk_from)                              */
    uint64_t tag;          /* EV_DELIVER tag: u64 in the reference (endpoint.rs:120)                 */
    uint32_t val;          /* EV_DELIVER payload                                                     */
    uint64_t aux;          /* typed RPC request payload: the caller's rsp_tag (rpc.rs:121-124)       */
    uint8_t  node;         /* EV_RESTART                                                             */
    uint8_t  is_rsp, hook_valid, hook_all, hook_code;   /* EV_DELIVER: the hooks_rsp entry of the destination node,
                              cloned when the message was sent and consulted when it arrives (net/mod.rs:321-328) */
} event_t;

typedef struct { uint64_t tag; uint16_t slot, gen; uint32_t rxseq; uint8_t tag8; } reg_t; /* (tag, oneshot::Sender).  rxseq names the oneshot channel: a counter that never
                                                                               repeats here (the device keeps 8 bits of it: reg_model_limits below); tag8 = the
                                                                               tag byte of the device's registration word: the tag, 0xff for an rsp_tag */
typedef struct { uint64_t tag; uint32_t from; uint32_t val; uint64_t aux; } msg_t;   /* endpoint.rs:288-292 */

typedef struct { uint32_t val; uint8_t has_arrive; uint64_t arrive; } cmsg_t;   /* (Payload, State) net/mod.rs:411-415 */
typedef struct {                          /* one direction of a connection: net/mod.rs:367-405 channel()            */
    uint8_t tx_alive, rx_alive;           /* PayloadSender / PayloadReceiver still held                              */
    VEC(cmsg_t) q;                        /* tokio unbounded mpsc                                                    */
    int32_t rx_task; uint16_t rx_gen;     /* task parked in rx.recv()                                                */
} cdir_t;
typedef struct {
    uint8_t alive;
    uint8_t c_node, s_node;               /* connect1's `node` and `dst_node`                                        */
    int16_t guard_sock[2];                /* the Endpoint whose Arc<BindGuard> this end's Sender + Receiver hold clones of
                                             (endpoint.rs:181-190,203-210): [0] the client's, from connect1 on; [1] the
                                             listener's, from accept1 on; -1 = none (a connection nobody accepted yet)  */
    uint8_t dst_kind, dst_ipnode, src_kind; uint16_t dst_port, src_port;   /* its `dst` and `src` SocketAddrs (addr_t fields) */
    cdir_t d[2];                          /* [0] client -> server, [1] server -> client                             */
} conn_t;

typedef struct {
    uint8_t bound; uint8_t gen;           /* gen: which Endpoint object currently owns the address   */
    uint16_t port;                        /* the port it is (or was last) bound to: the table entry's, or — entry port 0 —
                                             the ephemeral port Network::bind picked (network.rs:224-236)             */
    uint16_t owner_slot, owner_gen;
    uint8_t ep_alive;                     /* the Endpoint object exists (its task has not dropped it)                */
    uint16_t guards;                      /* live (Sender, Receiver) pairs holding clones of its Arc<BindGuard>: the address
                                             stays in the node's socket table until the Endpoint AND all of them are gone */
    VEC(reg_t) registered;                /* endpoint.rs:298-303 Mailbox                             */
    VEC(msg_t) msgs;
    VEC(uint8_t) acceptq;                 /* conn_tx/conn_rx: pending connections (endpoint.rs:18,307) */
    int32_t acc_task; uint16_t acc_gen;   /* task parked in accept1's conn_rx.recv()                  */
} sock_t;

enum { AW_NONE = 0 };

typedef struct {
    uint8_t  alive;        /* the future exists (spawned, not yet completed/dropped)                 */
    uint16_t gen;
    uint8_t  prog, node;
    uint8_t  info_gen;     /* which NodeInfo of its node this task holds                             */
    uint8_t  killed;       /* this task's Arc<NodeInfo>.killed                                       */
    uint8_t  cancelled;    /* TaskInfo.cancelled (task/mod.rs:84)                                    */
    uint8_t  scheduled, running;          /* async-task SCHEDULED / RUNNING bits            [DEP A.7] */
    uint16_t pc; uint8_t sub;
    uint64_t deadline;     /* the Sleep currently awaited (time/sleep.rs:21-24)                      */
    uint64_t deadline2;    /* the Sleep inside timeout()                                             */
    uint64_t t0;
    uint16_t cnt[2];
    uint32_t val; uint32_t from;
    uint64_t aux;          /* rsp_tag of the typed RPC request in hand (rpc.rs:163-165)             */
    uint64_t rsp_tag;      /* rsp_tag of the call in flight (rpc.rs:121)                            */
    uint8_t  inbox_full; uint32_t rxseq;  /* the oneshot::Receiver currently held                    */
    uint64_t inbox_aux;
    int32_t  joiner; uint16_t joiner_gen; /* async-task awaiter                                      */
    uint8_t  join_state;                  /* parked in MS_OP_JOIN: 1 awaiting, 2 / 3 the awaited task completed / was cancelled */
    int8_t   conn; uint8_t side;          /* the (Sender, Receiver) pair this task holds, and which end */
    uint32_t cval; uint8_t chas; uint64_t carrive; uint32_t backoff_ms;   /* receiver stream state (net/mod.rs:386-400) */
} task_t;

enum { H_NONE = 0, H_RUNNING = 1, H_COMPLETED = 2, H_CANCELLED = 3 };
typedef struct { uint8_t state; uint16_t slot, gen; } handle_t;

typedef struct { uint16_t slot, gen; } tref_t;
typedef struct {
    uint8_t killed, paused;               /* the CURRENT Arc<NodeInfo>'s flags (task/mod.rs:100-103)  */
    uint8_t info_gen;                     /* how many times Handle::restart replaced the NodeInfo     */
    uint8_t gen0_killed;                  /* the NodeInfo captured by NodeHandles at build() is dead  */
    /* NetSim.hooks_req / hooks_rsp entries of this node (net/mod.rs:250-284): valid, mode (1 = drop all), tag, code */
    uint8_t hreq_valid, hreq_all, hreq_tag, hreq_code, hrsp_valid, hrsp_all, hrsp_code;
    VEC(uint16_t) paused_list;            /* Node.paused: Vec<Runnable> (task/mod.rs:345-350)        */
    VEC(tref_t) tasks;                    /* NodeInfo.tasks: Vec<Weak<TaskInfo>> in spawn order (:105) */
} node_t;

typedef struct {
    const madsim_workload_t* w;
    /* GlobalRng */
    xoshiro_t rng; uint64_t rng_calls; int buggify;
    uint64_t trace_hash; uint8_t* log; uint64_t log_len, log_cap;
    int no_log;                          /* madsim_limits_t.no_trace_hash: the reference's plain run — `log` and `check` both None (rand.rs:67) */
    /* Clock + Timer */
    uint64_t clock;
    uint64_t base_time_ns; /* Clock.base_time since UNIX_EPOCH (time/mod.rs:26-33)                      */
    VEC(event_t) heap;
    /* Executor */
    VEC(uint16_t) ready;
    VEC(task_t) tasks;
    handle_t* handles;
    node_t* nodes;
    sock_t* socks;
    /* Network */
    uint64_t clog_in, clog_out;           /* HashSet<NodeId> as bit sets (network.rs:27-28)          */
    uint64_t* clog_link;                  /* [n_nodes+1] rows of bits (network.rs:29)                */
    uint64_t loss_pint; int loss_always;  /* Bernoulli p_int                                 [DEP A.4] */
    int lat_mode; uint64_t lat_low, lat_range, lat_zone; /* UniformDuration               [DEP A.3] */
    const madsim_config_t* cfg;
    /* accounting */
    uint8_t panic_code;                                  /* message code of the panic being unwound */
    uint32_t ipvs_rr[MADSIM_MAX_SERVICES];               /* Service.rr_index of every virtual service (net/ipvs.rs:37-41) */
    uint8_t ipvs_present[MADSIM_MAX_SERVICES];           /* the service is in the HashMap (ipvs.rs:11-13) */
    uint16_t ipvs_n[MADSIM_MAX_SERVICES];                /* Service.servers: Vec<String>, as socket-table entries of the addresses */
    uint8_t ipvs_srv[MADSIM_MAX_SERVICES][256];          /* (a Vec: the device's six-server capacity is not the oracle's concern) */
    uint64_t msg_count; uint32_t steps; uint64_t obs_hash;
    uint32_t greg[4];                     /* Arc<AtomicUsize> flags shared by the test's tasks       */
    VEC(conn_t) conns;
    uint32_t panic; int main_slot;
    int unsupported;                                     /* the seed left the workload model (MADSIM_UNSUPPORTED): stop at once */
    int model_limits;                                    /* 1: the workload MODEL's ceilings decide verdicts (what the device runner reports);
                                                            0: the reference's unbounded containers all the way, ceilings only recorded */
    uint32_t model_events;                               /* MADSIM_ORACLE_ME_* of this seed, recorded either way */
    madsim_oracle_stats_t st;
} sim_t;

/* ------------------------------------------------------------------------------------------------
 * The workload MODEL's ceilings — a layer on top of the restatement, not part of it.
 * The reference's containers are unbounded (Vec mailboxes, Vec<Weak<TaskInfo>>, unbounded channels); the device runner's workload
 * model is not: 254 live tasks, 255 registrations per socket, 8-bit receive-sequence / generation bytes in a registration word, 15
 * queued channel payloads, 8 connections waiting for accept1, 6 servers per IPVS service — and the table FORMAT cannot say some things
 * Rust can (two Endpoints under one port-0 entry, a formatted panic message beyond the values its patterns were evaluated for).
 * Every such event is recorded in sim_t.model_events.  With model_limits on (the default: what every parity test compares the
 * device with) the seed's verdict becomes MADSIM_UNSUPPORTED, decided at the same instruction as on the device; with it off
 * (madsim_oracle_run_batch_pure) nothing in the run depends on a ceiling — the containers simply grow — and the caller derives the
 * runner verdict from the event mask.  tests/test_oracle_model_limits.py: on every seed without an event the two runs are
 * byte-identical.  Returns whether the caller should stop the run at once (only ever with the limits on).
 * ---------------------------------------------------------------------------------------------- */
static int model_event(sim_t* S, uint32_t bit) {
    S->model_events |= bit;
    if (S->model_limits) S->unsupported = 1;
    return S->model_limits;
}

/* ------------------------------------------------------------------------------------------------
 * GlobalRng::with + determinism log                                   rand.rs:64-88, A.6
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t rng_next(sim_t* S) { S->rng_calls++; return oracle_xoshiro_next(S->rng.s); }

/* One call per GlobalRng::with(): v = clone.gen::<u8>() ^ xor-fold(elapsed.as_nanos() as u128).
 * gen::<u8>() = next_u32() as u8 [DEP rand 0.8 Standard]; next_u32 = (next_u64 >> 32) [DEP xoshiro]. */
static void rng_log(sim_t* S) {
    if (S->no_log) return;                                  /* rand.rs:67: neither logging nor checking */
    xoshiro_t c = S->rng;
    uint8_t v = (uint8_t)(oracle_xoshiro_next(c.s) >> 32);
    uint64_t t = S->clock;
    for (int i = 0; i < 8; i++) v ^= (uint8_t)(t >> (8 * i));
    S->trace_hash = (S->trace_hash ^ v) * FNV_PRIME;
    if (S->log && S->log_len < S->log_cap) S->log[S->log_len] = v;
    S->log_len++;
}

/* [DEP rand 0.8 UniformInt<u64>::sample_single_inclusive via gen_range(lo..hi)] A.2
 * call sites: utils/mpsc.rs:76, task/mod.rs:320, time/mod.rs:31, net/mod.rs:288,290 — each wrapped
 * in ONE GlobalRng::with, so one log byte regardless of rejections. */
static uint64_t gen_range_u64(sim_t* S, uint64_t lo, uint64_t hi) {
    uint64_t range = hi - lo;   /* (hi-1) - lo + 1 */
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    uint64_t res;
    for (;;) {
        uint64_t v = rng_next(S);
        unsigned __int128 m = (unsigned __int128)v * range;
        if ((uint64_t)m <= zone) { res = lo + (uint64_t)(m >> 64); break; }
    }
    rng_log(S);
    return res;
}

/* [DEP rand 0.8 Bernoulli] via GlobalRng's RngCore impl (rand.rs:142-158): one with() per draw. */
static int gen_bool_pint(sim_t* S, uint64_t p_int, int always) {
    if (always) return 1;
    uint64_t v = rng_next(S);
    rng_log(S);
    return v < p_int;
}

/* [DEP rand 0.8 UniformDuration::new(lo,hi).sample] A.3.  `direct`: called on GlobalRng through its
 * RngCore impl (network.rs:267) => one with() per attempt; otherwise inside one with(). */
void oracle_uniform_duration_params(uint64_t lo, uint64_t hi, int* mode, uint64_t* low,
                                    uint64_t* range, uint64_t* zone) {
    uint64_t h = hi - 1;
    uint64_t lo_s = lo / NS_PER_S, lo_n = lo % NS_PER_S, hi_s = h / NS_PER_S, hi_n = h % NS_PER_S;
    if (hi_n < lo_n) { hi_s -= 1; hi_n += NS_PER_S; }
    if (lo_s == hi_s) {            /* Small: Uniform<u32>::new_inclusive(lo_n, hi_n) */
        uint32_t r = (uint32_t)(hi_n - lo_n + 1);
        uint32_t reject = r ? (uint32_t)((0xffffffffu - r + 1u) % r) : 0;
        *mode = 0; *low = lo_s * NS_PER_S + lo_n; *range = r; *zone = 0xffffffffu - reject;
    } else {                       /* Medium: Uniform<u64>::new_inclusive(lo, hi-1) */
        uint64_t r = h - lo + 1;
        uint64_t reject = r ? (UINT64_MAX - r + 1) % r : 0;
        *mode = 1; *low = lo; *range = r; *zone = UINT64_MAX - reject;
    }
}

static uint64_t sample_duration(sim_t* S, int mode, uint64_t low, uint64_t range, uint64_t zone,
                                int direct) {
    uint64_t res;
    for (;;) {
        uint64_t v = rng_next(S);
        if (direct) rng_log(S);
        if (mode == 0) {
            uint64_t m = (uint64_t)(uint32_t)(v >> 32) * (uint64_t)(uint32_t)range; /* next_u32 */
            if ((uint32_t)m <= (uint32_t)zone) { res = low + (m >> 32); break; }
        } else {
            unsigned __int128 m = (unsigned __int128)v * range;
            if ((uint64_t)m <= zone) { res = low + (uint64_t)(m >> 64); break; }
        }
    }
    if (!direct) rng_log(S);
    return res;
}

/* ------------------------------------------------------------------------------------------------
 * Timer = BinaryHeap<Event>, Ord reversed on deadline only            [DEP A.5]
 * Rust alloc::collections::BinaryHeap push / pop / sift_up / sift_down_to_bottom restated on the
 * array.  "a <= b" in heap order means a.deadline >= b.deadline.
 * ---------------------------------------------------------------------------------------------- */
static void heap_sift_up(event_t* d, size_t start, size_t pos) {
    event_t hole = d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (hole.deadline >= d[parent].deadline) break;   /* hole <= parent: stop */
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = hole;
}

static void timer_add(sim_t* S, event_t e) {              /* time/mod.rs:158-165 -> Timer::add */
    vec_push(S->heap, e);
    heap_sift_up(S->heap.p, 0, S->heap.n - 1);
    if (S->heap.n > S->st.max_heap) S->st.max_heap = (uint32_t)S->heap.n;
}

static event_t timer_pop(sim_t* S) {                      /* BinaryHeap::pop */
    event_t* d = S->heap.p;
    event_t item = d[--S->heap.n];
    if (S->heap.n > 0) {
        event_t top = d[0]; d[0] = item; item = top;
        size_t end = S->heap.n, pos = 0;
        event_t hole = d[0];
        size_t child = 1;
        while (child + 1 < end) {                         /* child <= end.saturating_sub(2) */
            if (d[child].deadline >= d[child + 1].deadline) child++;  /* left <= right: take right */
            d[pos] = d[child]; pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1) { d[pos] = d[child]; pos = child; }
        d[pos] = hole;
        heap_sift_up(d, 0, pos);
    }
    return item;
}

/* ------------------------------------------------------------------------------------------------
 * async-task wake / schedule                                          [DEP A.7], task/mod.rs:640-651
 * ---------------------------------------------------------------------------------------------- */
static void ready_push(sim_t* S, uint16_t slot) {         /* utils/mpsc.rs:55-61 Vec::push */
    vec_push(S->ready, slot);
    if (S->ready.n > S->st.max_ready) S->st.max_ready = (uint32_t)S->ready.n;
}

static void wake(sim_t* S, uint16_t slot, uint16_t gen) {
    if (slot >= S->tasks.n) return;
    task_t* t = &S->tasks.p[slot];
    if (!t->alive || t->gen != gen) return;               /* COMPLETED | CLOSED: no-op */
    if (t->scheduled) return;
    t->scheduled = 1;
    if (!t->running) ready_push(S, slot);                 /* RUNNING: run() re-queues after the poll */
}

/* ------------------------------------------------------------------------------------------------
 * Network                                                             net/network.rs
 * ---------------------------------------------------------------------------------------------- */
static int link_clogged(sim_t* S, unsigned src, unsigned dst) {        /* network.rs:199-203 */
    return ((S->clog_out >> src) & 1) || ((S->clog_in >> dst) & 1) || ((S->clog_link[src] >> dst) & 1);
}

/* A SocketAddr: `kind` = MADSIM_ADDR_* picks the IP (10.0.0.<node> / 0.0.0.0 / 127.0.0.1). */
typedef struct { uint8_t kind, node; uint16_t port; } addr_t;
static addr_t addr_of_sock(const sim_t* S, unsigned idx) {       /* local_addr() of the Endpoint entry idx stands for */
    addr_t a = { S->w->socks[idx].kind, S->w->socks[idx].node, S->socks[idx].port }; return a;
}
/* The source address a receiver was shown, kept as  socket index | dst-was-loopback << 6 | port << 8  (network.rs:307-311):
 * the sender's real IP, or 127.0.0.1 when the datagram was addressed to a loopback address, with the port the sending
 * socket had at that moment. */
static uint32_t mk_from(const sim_t* S, unsigned idx, unsigned lb) { return idx | (lb << 6) | ((uint32_t)S->socks[idx].port << 8); }
static addr_t addr_of_from(const sim_t* S, uint32_t from) {
    addr_t a = { (uint8_t)((from & 0x40) ? MADSIM_ADDR_LOOPBACK : MADSIM_ADDR_IP), S->w->socks[from & 0x3f].node, (uint16_t)(from >> 8) };
    return a;
}
static int addr_eq(addr_t x, addr_t y) {                  /* SocketAddr equality */
    return x.kind == y.kind && x.port == y.port && ((x.kind != MADSIM_ADDR_IP && x.kind != MADSIM_ADDR_VIRTUAL) || x.node == y.node);
}
static int node_has_ip(const sim_t* S, unsigned node) { return !(S->w->nodes[node].flags & MADSIM_NODE_NO_IP); }

/* node.sockets.get(&(addr, protocol)) on node `on` (network.rs:206-251 keys sockets by the address they were bound to). */
static int find_exact(sim_t* S, unsigned on, addr_t a) {
    for (uint32_t i = 0; i < S->w->n_socks; i++) {
        const madsim_sock_t* e = &S->w->socks[i];
        if (S->socks[i].bound && e->node == on && e->kind == a.kind && S->socks[i].port == a.port && (a.kind != MADSIM_ADDR_IP || a.node == on)) return (int)i;
    }
    return -1;
}

/* Network::resolve_dest_node (network.rs:272-290): the node a datagram for `dst` goes to, or -1 (dropped, no draws). */
static int resolve_dest_node(sim_t* S, unsigned node, addr_t dst) {
    if (dst.kind == MADSIM_ADDR_LOOPBACK || find_exact(S, node, dst) >= 0) return (int)node;
    if (!node_has_ip(S, node)) return -1;                                  /* "ip not set" */
    if (dst.kind == MADSIM_ADDR_IP && dst.node >= 1 && dst.node <= S->w->n_nodes && node_has_ip(S, dst.node)) return dst.node;   /* addr_to_node */
    return -1;                                                             /* "destination not found" */
}

/* `if let Some(addr) = self.ipvs.get_server(ServiceAddr::from_addr_proto(dst, protocol)) { dst = addr.parse() }`
 * (net/mod.rs:312-317 in send, :345-350 in connect1) with IpVirtualServer::get_server (net/ipvs.rs:88-105): the service whose
 * address equals `dst`; no servers -> None; `if *i >= len { *i = 0 }; server = servers[*i]; *i += 1`. */
static addr_t ipvs_rewrite(sim_t* S, addr_t dst) {
    const madsim_workload_t* w = S->w;
    for (uint32_t k = 0; k < w->n_services; k++) {
        const madsim_service_t* sv = &w->services[k];
        const addr_t va = { w->socks[sv->vaddr].kind, w->socks[sv->vaddr].node, w->socks[sv->vaddr].port };
        if (!addr_eq(va, dst)) continue;
        if (!S->ipvs_present[k]) return dst;              /* services.get_mut(..)? : None */
        if (S->ipvs_n[k] == 0) return dst;                /* Some(service) with no servers: None */
        uint32_t* i = &S->ipvs_rr[k];
        if (*i >= S->ipvs_n[k]) *i = 0;
        const madsim_sock_t* e = &w->socks[S->ipvs_srv[k][*i]];
        *i += 1;
        addr_t real = { e->kind, e->node, e->port };
        return real;
    }
    return dst;
}

/* Network::try_send (network.rs:296-313) + test_link (:261-269).  Returns 1 and the latency / socket / the `from`
 * flag when a delivery must be scheduled, 0 when the message is dropped, -1 when the sender's task panics
 * (`.ip.unwrap()` of an IP-less node, :309). */
static int try_send(sim_t* S, unsigned src_node, addr_t dst, uint64_t* latency, int* dst_sock, unsigned* from_lb) {
    int dst_node = resolve_dest_node(S, src_node, dst);
    if (dst_node < 0) return 0;                           /* no draw */
    if (link_clogged(S, src_node, (unsigned)dst_node)) return 0;    /* no draw */
    if (gen_bool_pint(S, S->loss_pint, S->loss_always)) return 0;
    S->msg_count++;
    *latency = sample_duration(S, S->lat_mode, S->lat_low, S->lat_range, S->lat_zone, 1);
    int s = find_exact(S, (unsigned)dst_node, dst);                        /* sockets.get(&(dst, protocol)) */
    if (s < 0) { addr_t any = { MADSIM_ADDR_UNSPECIFIED, 0, dst.port }; s = find_exact(S, (unsigned)dst_node, any); }   /* .or_else(0.0.0.0:port) */
    if (s < 0) return 0;                                  /* draws consumed, silently dropped */
    *from_lb = dst.kind == MADSIM_ADDR_LOOPBACK;
    if (!*from_lb && !node_has_ip(S, src_node)) return -1;
    *dst_sock = s;
    return 1;
}

/* Mailbox::deliver (endpoint.rs:331-351) through EndpointSocket::deliver (:311-318). */
static void mailbox_deliver(sim_t* S, event_t* e) {
    sock_t* k = &S->socks[e->sock];
    if (!k->bound || k->gen != e->sockgen) return;        /* that EndpointSocket has left the table */
    size_t i = 0;
    while (i < k->registered.n) {
        if (k->registered.p[i].tag == e->tag) {
            reg_t r = k->registered.p[i];
            k->registered.p[i] = k->registered.p[--k->registered.n];     /* swap_remove */
            task_t* t = r.slot < S->tasks.n ? &S->tasks.p[r.slot] : NULL;
            if (t && t->alive && t->gen == r.gen && t->rxseq == r.rxseq && !t->inbox_full) {
                t->inbox_full = 1; t->val = e->val; t->from = e->from; t->inbox_aux = e->aux;   /* oneshot send Ok */
                wake(S, r.slot, r.gen);                                   /* [DEP tokio oneshot] */
                return;
            }
            /* receiver dropped: try next (i stays) */
        } else {
            i++;
        }
    }
    if (!k->ep_alive) return;     /* the Endpoint object is gone (the address is held by its connections, or its node was killed before the
                                     guard dropped: net/mod.rs:483-493): a receive registered by another holder was served above; with
                                     none, nobody is left to read what would be queued here */
    if (k->msgs.n >= MADSIM_MAX_MBOX_MSGS) model_event(S, MADSIM_ORACLE_ME_MSGS);     /* the model's ceiling: a 256th queued message */
    msg_t m = { e->tag, e->from, e->val, e->aux };
    vec_push(k->msgs, m);
    if (k->msgs.n > S->st.max_msgs) S->st.max_msgs = (uint32_t)k->msgs.n;
}

/* ------------------------------------------------------------------------------------------------
 * task lifecycle
 * ---------------------------------------------------------------------------------------------- */
static void sock_drop_acceptq(sim_t* S, sock_t* k);
/* A registration about to be pushed.  The workload model identifies a registration's receiver by 8 bits of its receive sequence number and 8 bits of
 * its task generation (the device's registration word): a DEAD registration still in the list that agrees with the new one in tag, slot and both
 * low bytes would be taken for it — possible only after one of the two counters has wrapped (256 receives of one task with a dead registration
 * surviving, 256 instances of one slot).  Such a seed leaves the model (MADSIM_UNSUPPORTED; the device checks the same thing, k_poll.h
 * may_have_twin); it also enforces the model's ceiling of registrations per socket.  The run goes on: the verdict is the whole answer. */
static void reg_model_limits(sim_t* S, const sock_t* k, const reg_t* r) {
    if (k->registered.n >= MADSIM_MAX_MBOX_REGS) model_event(S, MADSIM_ORACLE_ME_REGS);
    for (size_t i = 0; i < k->registered.n; i++) {
        const reg_t* o = &k->registered.p[i];
        if (o->tag8 == r->tag8 && o->slot == r->slot && (o->gen & 0xff) == (r->gen & 0xff) && (o->rxseq & 0xff) == (r->rxseq & 0xff))
            model_event(S, MADSIM_ORACLE_ME_REG_ALIAS);
    }
}
/* An EndpointSocket lives while the node's socket table (`bound`) or its Endpoint (`ep_alive`) holds an Arc of it (in-flight
 * delivery closures are not counted); when it dies conn_tx dies, and with it the connections nobody accepted. */
static void sock_maybe_free(sim_t* S, sock_t* k) {
    if (!k->bound && !k->ep_alive && k->acceptq.n) sock_drop_acceptq(S, k);
}
/* The address leaves the node's socket table — Network::close (network.rs:253-258) from BindGuard::drop. */
static void sock_release(sim_t* S, sock_t* k) {
    k->bound = 0;
    sock_maybe_free(S, k);
}
/* drop(Endpoint): conn_rx goes (the async_channel is closed: later connections are dropped on arrival, endpoint.rs:320-328)
 * and the Arc<BindGuard> loses one owner.  BindGuard::drop (net/mod.rs:483-493) runs with the LAST owner and does nothing
 * when the binder's NodeInfo is killed: after Handle::kill reset_node has emptied the socket table already, after the init
 * task's exit (Spawner::exit is info.kill() alone, task/mod.rs:657-661) the address simply stays in the table. */
static void endpoint_drop(sim_t* S, sock_t* k, int node_killed) {
    k->ep_alive = 0; k->acc_task = -1;
    if (!node_killed && k->guards == 0) k->bound = 0;
    sock_maybe_free(S, k);
}
/* one end's Sender and Receiver are dropped: their two clones of the Endpoint's Arc<BindGuard> */
static void guard_release(sim_t* S, int sock, int node_killed) {
    if (sock < 0 || node_killed) return;                  /* killed: reset_node forgot the counts, drop() would return early */
    sock_t* k = &S->socks[sock];
    if (k->guards && --k->guards == 0 && !k->ep_alive && k->bound) sock_release(S, k);
}
static void sock_close_owned(sim_t* S, uint16_t slot, uint16_t gen, int node_killed) {
    for (uint32_t i = 0; i < S->w->n_socks; i++) {
        sock_t* k = &S->socks[i];
        if (k->ep_alive && k->owner_slot == slot && k->owner_gen == gen) endpoint_drop(S, k, node_killed);
    }
}

/* `via_handle`: spawned through a NodeHandle captured at build() (NodeHandle::spawn from another node's
 * task: the Spawner holds the ORIGINAL Arc<NodeInfo>); otherwise task::spawn / init on the current info. */
static int spawn_task_from(sim_t* S, unsigned prog, int record_handle, int via_handle, int parent);
static int spawn_task(sim_t* S, unsigned prog, int record_handle) { return spawn_task_from(S, prog, record_handle, 0, -1); }
/* parent >= 0: task::spawn from inside that task's context — its own Arc<NodeInfo>, whatever became of the node since
 * (Spawner::current -> context::current_task().node: a guard's Drop running while its killed task is dropped) */
static int spawn_task_from(sim_t* S, unsigned prog, int record_handle, int via_handle, int parent) {   /* task/mod.rs:627-654 */
    size_t slot = 0;
    while (slot < S->tasks.n && S->tasks.p[slot].alive) slot++;
    /* the reference's task set is unbounded; the workload model holds MADSIM_MAX_LIVE_TASKS live tasks (the device's 8-bit task slot, filled
     * lowest free slot first like this Vec): a 255th leaves the model at this spawn — MADSIM_UNSUPPORTED, reported when the run ends (this
     * restatement simply goes on; the verdict is the whole answer, include/madsim_hip.h) */
    if (slot >= MADSIM_MAX_LIVE_TASKS) model_event(S, MADSIM_ORACLE_ME_TASKS);
    if (slot == S->tasks.n) { task_t z; memset(&z, 0, sizeof z); vec_push(S->tasks, z); }
    task_t* t = &S->tasks.p[slot];
    uint16_t gen = (uint16_t)(t->gen + 1);
    memset(t, 0, sizeof *t);
    t->alive = 1; t->gen = gen; t->prog = (uint8_t)prog; t->node = S->w->progs[prog].node;
    {
        node_t* n = &S->nodes[t->node];
        if (parent >= 0) { t->killed = S->tasks.p[parent].killed; t->info_gen = S->tasks.p[parent].info_gen; }
        else if (via_handle && n->info_gen != 0) { t->killed = 1; t->info_gen = 0; }        /* stale handle: dead info */
        else { t->killed = via_handle ? n->gen0_killed : n->killed; t->info_gen = n->info_gen; }  /* :632-634 */
        if (t->info_gen == n->info_gen) { tref_t r = { (uint16_t)slot, gen }; vec_push(n->tasks, r); }
    }
    t->pc = S->w->progs[prog].entry; t->joiner = -1; t->conn = -1;
    /* `from` before the task has received anything: no Rust program can name it (the binding does not exist yet); the workload VM
     * can (MS_OP_REPLY after a timed-out receive), and then both this oracle and the kernel read it as socket-table entry 0 */
    if (S->w->n_socks) t->from = (uint32_t)S->w->socks[0].port << 8;
    t->scheduled = 1;                                     /* runnable.schedule() :651 */
    ready_push(S, (uint16_t)slot);
    if (record_handle) { S->handles[prog].state = H_RUNNING; S->handles[prog].slot = (uint16_t)slot; S->handles[prog].gen = gen; }
    uint32_t live = 0; for (size_t i = 0; i < S->tasks.n; i++) live += S->tasks.p[i].alive;
    if (live > S->st.max_tasks) S->st.max_tasks = live;
    return (int)slot;
}


/* NodeInfo::kill (task/mod.rs:133-140): flag + wake every task of that info, in spawn order. */
static void info_kill(sim_t* S, unsigned node) {
    node_t* n = &S->nodes[node];
    n->killed = 1;
    if (n->info_gen == 0) n->gen0_killed = 1;
    size_t cnt = n->tasks.n;
    tref_t* list = n->tasks.p;
    n->tasks.p = NULL; n->tasks.n = n->tasks.cap = 0;       /* drain(..) */
    for (size_t i = 0; i < cnt; i++) {
        task_t* t = list[i].slot < S->tasks.n ? &S->tasks.p[list[i].slot] : NULL;
        if (t && t->alive && t->gen == list[i].gen) { t->killed = 1; wake(S, list[i].slot, list[i].gen); }
    }
    free(list);
}

static void task_finish(sim_t* S, uint16_t slot, int outcome);
static void task_finish_opt(sim_t* S, uint16_t slot, int outcome, int guard);

static void paused_clear(sim_t* S, unsigned node) {       /* node.paused.clear(): drops the Runnables */
    node_t* n = &S->nodes[node];
    size_t cnt = n->paused_list.n;
    uint16_t* list = n->paused_list.p;
    n->paused_list.p = NULL; n->paused_list.n = n->paused_list.cap = 0;
    for (size_t i = 0; i < cnt; i++) task_finish(S, list[i], H_CANCELLED);
    free(list);
}

static void node_kill(sim_t* S, unsigned node) {          /* TaskHandle::kill_id (task/mod.rs:362-371) */
    paused_clear(S, node);
    info_kill(S, node);
    for (uint32_t i = 0; i < S->w->n_socks; i++)          /* NetSim::reset_node -> sockets.clear() (network.rs:142-147) */
        if (S->w->socks[i].node == node) { S->socks[i].bound = 0; S->socks[i].guards = 0; sock_maybe_free(S, &S->socks[i]); }
}

static void node_restart(sim_t* S, unsigned node) {       /* TaskHandle::restart (task/mod.rs:374-401) */
    node_t* n = &S->nodes[node];
    size_t cnt = n->tasks.n; tref_t* old = n->tasks.p;    /* old_info keeps its own task list */
    n->tasks.p = NULL; n->tasks.n = n->tasks.cap = 0;
    if (n->info_gen == 0) n->gen0_killed = 1;
    n->info_gen++; n->killed = 0; n->paused = 0;          /* new_info */
    paused_clear(S, node);
    for (size_t i = 0; i < cnt; i++) {                    /* old_info.kill() */
        task_t* t = old[i].slot < S->tasks.n ? &S->tasks.p[old[i].slot] : NULL;
        if (t && t->alive && t->gen == old[i].gen) { t->killed = 1; wake(S, old[i].slot, old[i].gen); }
    }
    free(old);
    for (uint32_t p = 1; p < S->w->n_progs; p++)          /* init(&Spawner { new info }) */
        if (S->w->progs[p].node == node && (S->w->progs[p].flags & MADSIM_PROG_INIT)) spawn_task(S, p, 0);
}

/* ---- reliable channel: NetSim::connect1 / channel (net/mod.rs:337-430), Endpoint::accept1 (endpoint.rs:197-211) ---- */
/* the `test_link` closure of channel(): try_send(..).map(|latency| now + latency) (net/mod.rs:375-380) */
static int chan_test_link(sim_t* S, conn_t* c, int dir, uint64_t* arrive) {
    /* (tx1, rx1) = channel(node, dst), (tx2, rx2) = channel(dst_node, src)  (net/mod.rs:356-357) */
    unsigned src_node = dir == 0 ? c->c_node : c->s_node;
    addr_t to = { dir == 0 ? c->dst_kind : c->src_kind, dir == 0 ? c->dst_ipnode : c->c_node, dir == 0 ? c->dst_port : c->src_port };
    uint64_t lat; int ds; unsigned lb;
    const int sent = try_send(S, src_node, to, &lat, &ds, &lb);
    if (sent <= 0) return sent;
    *arrive = S->clock + lat;
    return 1;
}

static void conn_drop_handles(sim_t* S, int id, int side, int node_killed);
/* the EndpointSocket is freed: the connections still queued in conn_tx go with it (their server-side raw handles) */
static void sock_drop_acceptq(sim_t* S, sock_t* k) {
    size_t n = k->acceptq.n; k->acceptq.n = 0; k->acc_task = -1;
    for (size_t i = 0; i < n; i++) conn_drop_handles(S, k->acceptq.p[i], 1, 0);
}

/* drop(tx); drop(rx) of one end, by a task whose NodeInfo is killed or not */
static void conn_drop_handles(sim_t* S, int id, int side, int node_killed) {
    if (id < 0) return;
    conn_t* c = &S->conns.p[id];
    cdir_t* out = &c->d[side], *in = &c->d[1 - side];
    if (out->tx_alive) {
        out->tx_alive = 0;                                    /* last mpsc sender dropped: a parked receiver wakes */
        if (out->rx_task >= 0) { int32_t r = out->rx_task; out->rx_task = -1; wake(S, (uint16_t)r, out->rx_gen); }
    }
    in->rx_alive = 0; in->rx_task = -1;
    const int gs = c->guard_sock[side];
    c->guard_sock[side] = -1;
    if (!c->d[0].tx_alive && !c->d[0].rx_alive && !c->d[1].tx_alive && !c->d[1].rx_alive) {
        c->alive = 0; c->d[0].q.n = 0; c->d[1].q.n = 0;
    }
    guard_release(S, gs, node_killed);                    /* may free a socket and, with it, connections queued there */
}

/* The future is gone (completed, or dropped by the executor).  outcome: H_COMPLETED / H_CANCELLED. */
/* MADSIM_PROG_DROP_SPAWN: the guard moved into the body drops after its other locals: A::drop -> task::spawn in this task's
 * context (task/mod.rs:1190-1196, 1227-1233) */
static void task_drop_guard(sim_t* S, uint16_t slot) {
    const unsigned prog = S->tasks.p[slot].prog;
    if (S->w->progs[prog].flags & MADSIM_PROG_DROP_SPAWN) spawn_task_from(S, prog + 1u, 0, 0, slot);
}
static void task_finish(sim_t* S, uint16_t slot, int outcome) { task_finish_opt(S, slot, outcome, 1); }
static void task_finish_opt(sim_t* S, uint16_t slot, int outcome, int guard) {
    task_t* t = &S->tasks.p[slot];
    if (t->conn >= 0) { int id = t->conn; t->conn = -1; conn_drop_handles(S, id, t->side, t->killed); t = &S->tasks.p[slot]; }
    sock_close_owned(S, slot, t->gen, t->killed);
    if (guard) { task_drop_guard(S, slot); t = &S->tasks.p[slot]; }
    handle_t* h = &S->handles[t->prog];
    if (h->state == H_RUNNING && h->slot == slot && h->gen == t->gen) h->state = (uint8_t)outcome;
    int32_t j = t->joiner; uint16_t jg = t->joiner_gen;
    t->alive = 0; t->scheduled = 0; t->running = 0;
    if (j >= 0) {                                         /* async-task hands the output over and notifies the awaiter */
        task_t* jt = (size_t)j < S->tasks.n ? &S->tasks.p[j] : NULL;
        if (jt && jt->alive && jt->gen == jg && jt->join_state == 1) jt->join_state = outcome == H_CANCELLED ? 3 : 2;
        wake(S, (uint16_t)j, jg);
    }
}

/* ------------------------------------------------------------------------------------------------
 * one poll of a task's future (Runnable::run, task/mod.rs:279-283).  Returns 1 if the task panicked.
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t insn_dur(const madsim_insn_t* in) { return (uint64_t)in->b * NS_PER_S + in->imm; }

/* Sleep::poll (time/sleep.rs:47-54): Ready when elapsed, else register ANOTHER timer. */
static int sleep_poll(sim_t* S, uint16_t slot, uint64_t deadline) {
    if (S->clock >= deadline) return 1;
    event_t e; memset(&e, 0, sizeof e);
    e.deadline = deadline; e.kind = EV_WAKE; e.slot = slot; e.gen = S->tasks.p[slot].gen;
    timer_add(S, e);
    return 0;
}

/* TimeHandle::sleep / sleep_until (time/mod.rs:111-124): 1 ms floor. */
static uint64_t sleep_deadline(sim_t* S, uint64_t deadline) {
    uint64_t min_deadline = S->clock + NS_PER_MS;
    return deadline > min_deadline ? deadline : min_deadline;
}

/* NetSim::rand_delay first half (net/mod.rs:287-292): draws, returns the Sleep deadline. */
static uint64_t rand_delay_start(sim_t* S) {
    uint64_t delay = gen_range_u64(S, 0, 5) * 1000ull;
    if (S->buggify && gen_bool_pint(S, (uint64_t)(0.1 * 18446744073709551616.0), 0))
        delay = gen_range_u64(S, 1, 5) * NS_PER_S;
    return sleep_deadline(S, S->clock + delay);
}

static void timer_expire(sim_t* S, uint64_t now);

static int poll_task(sim_t* S, uint16_t slot) {
    const madsim_workload_t* w = S->w;
    for (;;) {
        task_t* t = &S->tasks.p[slot];
        if (t->pc >= w->n_insns) return 1;
        const madsim_insn_t* in = &w->insns[t->pc];
        switch (in->op) {                                  /* ops whose `a` names an Endpoint, through a port-0 table entry: */
        case MS_OP_SEND: case MS_OP_CONNECT: case MS_OP_RPC_CALL: case MS_OP_REPLY: case MS_OP_RECV:
        case MS_OP_RECV_TIMEOUT: case MS_OP_ACCEPT: case MS_OP_RPC_REPLY:
            /* the entry must still name the socket its last bind put in the table.  No Rust program uses an Endpoint it never
             * bound or has dropped; the table format can say it, and the device (whose port-0 entries of one (node, IP) share
             * their candidate sockets, geometry.h) could only answer with a stranger's mailbox: outside the model, on both sides
             * (`drop(ep)` of such a name stays a no-op: MS_OP_CLOSE below) */
            /* (model limits off: the op goes on against this entry's own — dead — socket, which is what this restatement's tables say) */
            if (in->a < w->n_socks && w->socks[in->a].port == 0 && !S->socks[in->a].bound && model_event(S, MADSIM_ORACLE_ME_EPH_STALE)) return 1;
            break;
        default: break;
        }
        switch (in->op) {
        case MS_OP_DONE:
            /* an init task is `async move { future.await; h.exit() }` (runtime/mod.rs:362-370): the body's locals — its
             * (tx, rx), its Endpoints — are gone when `future.await` returns, on a node that is not killed yet; then
             * Spawner::exit = info.kill() on the NodeInfo it was spawned with (task/mod.rs:657-661) */
            if ((w->progs[t->prog].flags & MADSIM_PROG_INIT) && t->info_gen == S->nodes[t->node].info_gen) {
                if (t->conn >= 0) { int id = t->conn; t->conn = -1; conn_drop_handles(S, id, t->side, 0); t = &S->tasks.p[slot]; }
                sock_close_owned(S, slot, t->gen, 0);
                task_drop_guard(S, slot);
                t = &S->tasks.p[slot];
                info_kill(S, t->node);
                task_finish_opt(S, slot, H_COMPLETED, 0);
                return 0;
            }
            task_finish(S, slot, H_COMPLETED);
            return 0;
        case MS_OP_SPAWN:
            {
                /* another node's program: NodeHandle::spawn (the handle's ORIGINAL NodeInfo); this node's: task::spawn =
                 * Spawner::current() = this task's OWN Arc<NodeInfo> (task/mod.rs:592-599) — not the node's current one: a
                 * task that restarted its own node spawns under the dead incarnation */
                const int via = w->progs[in->a].node != t->node;
                int child = spawn_task_from(S, in->a, 1, via, via ? -1 : (int)slot);
                t = &S->tasks.p[slot];
                if ((in->b & 2) && child >= 0) {           /* `spawn(async move { .. tx, rx .. })`: the handles move */
                    S->tasks.p[child].conn = t->conn; S->tasks.p[child].side = t->side; t->conn = -1;
                }
                if ((in->b & MADSIM_SPAWN_MOVE_REQUEST) && child >= 0) {   /* rpc.rs:170: `spawn(async move { .. from, rsp_tag .. })` */
                    S->tasks.p[child].val = t->val; S->tasks.p[child].from = t->from; S->tasks.p[child].aux = t->aux;
                }
            }
            t->pc++;
            break;
        case MS_OP_BUILD:                                  /* create_node().init(..).build(): task/mod.rs:472-474 */
            for (uint32_t p = 1; p < w->n_progs; p++)
                if (w->progs[p].node == in->a && (w->progs[p].flags & MADSIM_PROG_INIT)
                    && !(w->progs[p].flags & MADSIM_PROG_PRE)) spawn_task(S, p, 0);
            t = &S->tasks.p[slot]; t->pc++;
            break;
        case MS_OP_JOIN: {                                 /* task/join.rs:59-72 + async-task poll_task */
            /* `handle.await` moves the JoinHandle into the await: the task it names when the await begins is the one awaited,
             * whatever a later spawn stores in handle[prog]; its outcome reaches the awaiter when it finishes (task_finish) */
            const int want_err = in->b & 1;
            int outcome;
            if (t->join_state == 1) return 0;                  /* woken for another reason (a stale timer): still pending */
            if (t->join_state) { outcome = t->join_state == 3 ? H_CANCELLED : H_COMPLETED; t->join_state = 0; }
            else {
                handle_t* h = &S->handles[in->a];
                if (h->state == H_RUNNING) {
                    task_t* c = &S->tasks.p[h->slot];
                    c->joiner = slot; c->joiner_gen = t->gen;  /* header.register(cx.waker()) */
                    t->join_state = 1;
                    return 0;
                }
                if (h->state == H_NONE) return 1;
                outcome = h->state;
            }
            if ((outcome == H_CANCELLED) != want_err) return 1;   /* unwrap()/unwrap_err() */
            t->pc++;
            break;
        }
        case MS_OP_ABORT: {                                /* AbortHandle::abort (task/join.rs:158-163) */
            handle_t* h = &S->handles[in->a];
            if (h->state == H_RUNNING) { S->tasks.p[h->slot].cancelled = 1; wake(S, h->slot, h->gen); }
            t = &S->tasks.p[slot]; t->pc++;
            break;
        }
        case MS_OP_KILL: node_kill(S, in->a); t = &S->tasks.p[slot]; t->pc++; break;
        case MS_OP_RESTART: node_restart(S, in->a); t = &S->tasks.p[slot]; t->pc++; break;
        case MS_OP_PAUSE: S->nodes[in->a].paused = 1; t->pc++; break;       /* task/mod.rs:404-410 */
        case MS_OP_RESUME: {                               /* task/mod.rs:413-424 */
            node_t* n = &S->nodes[in->a];
            n->paused = 0;
            for (size_t i = 0; i < n->paused_list.n; i++) ready_push(S, n->paused_list.p[i]);
            n->paused_list.n = 0;
            t->pc++;
            break;
        }
        case MS_OP_ASSERT_EXIT:                            /* Handle::is_exit (task/mod.rs:444-449) */
            if ((S->nodes[in->a].killed != 0) != ((in->b & 1) != 0)) return 1;
            t->pc++;
            break;
        case MS_OP_GSET: S->greg[in->a & 3] = in->imm; t->pc++; break;
        case MS_OP_GADD: S->greg[in->a & 3] += in->imm; t->pc++; break;
        case MS_OP_ASSERT_G: if (S->greg[in->a & 3] != in->imm) return 1; t->pc++; break;
        case MS_OP_PANIC_IF_G_LT: if (S->greg[in->a & 3] < in->imm) return 1; t->pc++; break;
        case MS_OP_YIELD:                                  /* [DEP tokio yield_now outside a runtime] */
            if (t->sub == 0) { t->sub = 1; wake(S, slot, t->gen); return 0; }
            t->sub = 0; t->pc++;
            break;
        case MS_OP_IPVS: {                                 /* IpVirtualServer::{add,del}_{service,server} (net/ipvs.rs:50-85) */
            const uint32_t k = in->b;
            if (in->a == MADSIM_IPVS_ADD_SERVICE) { S->ipvs_present[k] = 1; S->ipvs_n[k] = 0; S->ipvs_rr[k] = 0; }   /* insert: replaces */
            else if (in->a == MADSIM_IPVS_DEL_SERVICE) S->ipvs_present[k] = 0;
            else {
                if (!S->ipvs_present[k]) return 1;         /* .expect("service not found") */
                if (in->a == MADSIM_IPVS_ADD_SERVER) {
                    /* servers is an unbounded Vec (net/ipvs.rs:66-72); the workload model holds six servers per service (include/madsim_hip.h,
                     * MADSIM_UNSUPPORTED): a seventh leaves the model at this call, on the device at the same one */
                    if (S->ipvs_n[k] >= 6 && model_event(S, MADSIM_ORACLE_ME_IPVS)) return 1;
                    if (S->ipvs_n[k] < 256) S->ipvs_srv[k][S->ipvs_n[k]++] = (uint8_t)in->imm;      /* servers.push (a Vec: this array is simply long) */
                } else {                                    /* servers.retain(|addr| addr != server_addr): equal address strings */
                    const addr_t gone = addr_of_sock(S, in->imm);
                    uint32_t n = 0;
                    for (uint32_t j = 0; j < S->ipvs_n[k]; j++)
                        if (!addr_eq(addr_of_sock(S, S->ipvs_srv[k][j]), gone)) S->ipvs_srv[k][n++] = S->ipvs_srv[k][j];
                    S->ipvs_n[k] = (uint16_t)n;
                }
            }
            t->pc++; break;
        }
        case MS_OP_HOOK_REQ: {                             /* NetSim::hook_rpc_req (net/mod.rs:240-262): HashMap::insert */
            node_t* n = &S->nodes[in->a];
            n->hreq_valid = 1; n->hreq_all = in->b & 1; n->hreq_tag = (uint8_t)(in->b >> 8); n->hreq_code = (uint8_t)in->imm;
            t->pc++; break;
        }
        case MS_OP_HOOK_RSP: {                             /* NetSim::hook_rpc_rsp (net/mod.rs:264-284) */
            node_t* n = &S->nodes[in->a];
            n->hrsp_valid = 1; n->hrsp_all = in->b & 1; n->hrsp_code = (uint8_t)in->imm;
            t->pc++; break;
        }
        case MS_OP_PANIC:                                  /* the message code restart_on_panic_matching looks at */
            /* a run-time formatted message is the decimal text of its value: values beyond the workload's panic_dyn_max are outside
             * what the patterns were evaluated for */
            if (in->a & 1) {
                const uint32_t v = S->greg[in->b & 3] + in->imm, dyn_max = w->panic_dyn_max ? w->panic_dyn_max : 254u;
                /* the workload declared the largest value it formats: beyond it the patterns' verdict on the message is not in the table —
                 * outside the model; with the limits off the run goes on as for a message no pattern names */
                if (v > dyn_max) { model_event(S, MADSIM_ORACLE_ME_PANIC_DYN); S->panic_code = MADSIM_PANIC_CODE_OTHER; return 1; }
                S->panic_code = (uint8_t)v;
            } else S->panic_code = (uint8_t)in->imm;
            return 1;
        case MS_OP_SET:
            t->cnt[in->a & 1] = (uint16_t)in->imm; t->pc++;
            break;
        case MS_OP_DJNZ:
            if (--t->cnt[in->a & 1] != 0) t->pc = in->b; else t->pc++;
            break;
        case MS_OP_JMP:
            t->pc = in->b;
            break;
        case MS_OP_TRACE: {
            uint64_t v = in->imm + ((in->b & 1) ? t->cnt[in->a & 1] : 0);
            obs_record(v); S->obs_hash = (S->obs_hash ^ v) * FNV_PRIME; t->pc++;
            break;
        }
        case MS_OP_SLEEP:
        case MS_OP_SLEEP_UNTIL:
            if (t->sub == 0) {
                uint64_t base = in->op == MS_OP_SLEEP ? S->clock : t->t0;
                t->deadline = sleep_deadline(S, base + insn_dur(in));
                t->sub = 1;
            }
            if (!sleep_poll(S, slot, t->deadline)) return 0;
            t->sub = 0; t->pc++;
            break;
        case MS_OP_MARK:
            t->t0 = S->clock; t->pc++;
            break;
        case MS_OP_ASSERT_ELAPSED: {
            uint64_t el = S->clock - t->t0, d = insn_dur(in);
            int ok = in->a == 0 ? el == d : in->a == 1 ? el >= d : el < d;
            if (!ok) return 1;
            t->pc++;
            break;
        }
        case MS_OP_ADVANCE:                                /* time/mod.rs:103-106 */
            S->clock += insn_dur(in);
            t->pc++;
            timer_expire(S, S->clock);
            break;
        case MS_OP_BIND:                                   /* net/mod.rs:446-470, network.rs:206-251 */
            if (t->sub == 0) { t->deadline = rand_delay_start(S); t->sub = 1; }
            if (!sleep_poll(S, slot, t->deadline)) return 0;
            {
                const madsim_sock_t* a = &w->socks[in->a];
                uint32_t bind_err = 0;
                /* network.rs:215-222: a specified, non-loopback IP must be the node's own (an IP-less node takes any);
                 * table entries are per node for every kind, so binding another node's entry is "not available" too */
                uint16_t port = a->port;
                if (a->node != t->node) bind_err = MADSIM_VAL_ADDR_NOT_AVAILABLE;
                else if (port == 0) {                      /* :224-236 "resolve port if unspecified": the first free one */
                    /* a table entry names ONE Endpoint at a time: bound again while the Endpoint of its previous bind is alive, the
                     * two would coexist under one name — outside the workload model, and the verdict says so (MADSIM_UNSUPPORTED)
                     * (model limits off: the entry is bound again over its live Endpoint — this restatement's table has one socket per entry).
                     * KNOWN GAP: an entry re-bound while its previous address is only kept in the table by connections made
                     * from it (endpoint.rs:181-190) takes the next port, as in the reference — but this table, one socket per entry, then
                     * forgets the older address; a THIRD bind, or a connect1 to the forgotten address, would differ from the reference.
                     * The device keeps both addresses (geometry.h device_socks: two candidates per entry) and reports MADSIM_UNSUPPORTED when a
                     * third is needed; the event below (a port beyond the candidates) catches the cases this table can see. */
                    if (S->socks[in->a].bound && S->socks[in->a].ep_alive && model_event(S, MADSIM_ORACLE_ME_EPH_REBIND)) return 1;
                    addr_t cand = { a->kind, a->node, 0 };
                    for (uint32_t p = 1; p <= 65535 && port == 0; p++) { cand.port = (uint16_t)p; if (find_exact(S, t->node, cand) < 0) port = (uint16_t)p; }
                    if (port == 0) bind_err = MADSIM_VAL_ADDR_IN_USE;      /* "no available ephemeral port" */
                    else {
                        /* the workload model holds as many candidate ports per (node, IP) as the table has entries for it — twice that when
                         * connections can keep an address alive past its Endpoint (geometry.h device_socks): a port beyond them is outside it */
                        uint32_t cand_ports = 0; int conns = 0;
                        for (uint32_t j = 0; j < w->n_socks; j++) cand_ports += w->socks[j].node == a->node && w->socks[j].kind == a->kind;
                        for (uint32_t j = 0; j < w->n_insns; j++) conns |= w->insns[j].op == MS_OP_CONNECT;
                        if (port > cand_ports * (conns ? 2u : 1u)) model_event(S, MADSIM_ORACLE_ME_EPH_PORTS);
                    }
                } else {
                    addr_t want = { a->kind, a->node, port };
                    if (find_exact(S, t->node, want) >= 0) bind_err = MADSIM_VAL_ADDR_IN_USE;   /* :238-246 */
                }
                if (bind_err) {
                    if (!(in->b & 1)) return 1;            /* .unwrap() */
                    t->val = bind_err; t->sub = 0; t->pc++;
                    break;
                }
                if (in->b & 1) t->val = 0;
                if (in->b & 2) t->val = port;              /* ep.local_addr().unwrap().port() */
                sock_t* k = &S->socks[in->a];
                k->bound = 1; k->gen++; k->owner_slot = slot; k->owner_gen = t->gen; k->port = port;
                k->ep_alive = 1; k->guards = 0;
                k->registered.n = 0; k->msgs.n = 0;        /* a fresh Endpoint + Mailbox */
                k->acceptq.n = 0; k->acc_task = -1;
            }
            t->sub = 0; t->pc++;
            break;
        case MS_OP_SEND:
        case MS_OP_RPC_REPLY:                              /* rpc.rs:172-175: send_to_raw(from, rsp_tag, rsp) */
        case MS_OP_REPLY:                                  /* net/mod.rs:298-333 */
            if (t->sub == 0) { t->deadline = rand_delay_start(S); t->sub = 1; }
            if (!sleep_poll(S, slot, t->deadline)) return 0;
            {
                const addr_t dst = ipvs_rewrite(S, in->op == MS_OP_SEND ? addr_of_sock(S, in->b & 0xff) : addr_of_from(S, t->from));
                uint64_t lat; int ds; unsigned lb;
                const int sent = try_send(S, w->socks[in->a].node, dst, &lat, &ds, &lb);
                if (sent < 0) return 1;                    /* `.ip.unwrap()` on an IP-less node */
                if (sent) {
                    event_t e; memset(&e, 0, sizeof e);
                    e.deadline = S->clock + lat; e.kind = EV_DELIVER; e.sock = (uint8_t)ds;
                    e.sockgen = S->socks[ds].gen; e.from = mk_from(S, in->a, lb); e.tag = (uint8_t)(in->b >> 8);
                    e.val = in->imm;
                    if (in->op == MS_OP_RPC_REPLY) {
                        e.tag = t->aux; e.val = in->imm & 0xff;
                        const node_t* dn = &S->nodes[w->socks[ds].node];   /* hooks_rsp.get(&dst_node).cloned() (:321) */
                        e.is_rsp = 1; e.hook_valid = dn->hrsp_valid; e.hook_all = dn->hrsp_all; e.hook_code = dn->hrsp_code;
                    }
                    timer_add(S, e);
                }
            }
            t->sub = 0; t->pc++;
            break;
        case MS_OP_RECV: {                                 /* endpoint.rs:140-149, 353-362 */
            sock_t* k = &S->socks[in->a];
            uint8_t tag = (uint8_t)(in->b >> 8);
            if (t->sub == 0) {
                t->rxseq++; t->inbox_full = 0;
                size_t idx = 0;
                while (idx < k->msgs.n && k->msgs.p[idx].tag != tag) idx++;
                if (idx < k->msgs.n) {
                    msg_t m = k->msgs.p[idx];
                    k->msgs.p[idx] = k->msgs.p[--k->msgs.n];            /* swap_remove */
                    t->inbox_full = 1; t->val = m.val; t->from = m.from; t->inbox_aux = m.aux;
                } else {
                    reg_t r = { tag, slot, t->gen, t->rxseq, (uint8_t)tag };
                    reg_model_limits(S, k, &r);
                    vec_push(k->registered, r);
                    if (k->registered.n > S->st.max_regs) S->st.max_regs = (uint32_t)k->registered.n;
                }
                t->sub = 1;
            }
            if (t->sub == 1) {
                if (!t->inbox_full) return 0;              /* oneshot rx Pending */
                t->inbox_full = 0;
                if (tag >= MADSIM_TAG_RPC_FIRST) t->aux = t->inbox_aux;   /* (rsp_tag, req, data) = *data.downcast() */
                t->deadline = rand_delay_start(S); t->sub = 2;
            }
            if (!sleep_poll(S, slot, t->deadline)) return 0;
            t->sub = 0; t->pc++;
            break;
        }
        case MS_OP_CONNECT:                                /* Endpoint::connect1 -> NetSim::connect1 (net/mod.rs:337-364) */
            if (t->sub == 0) { t->deadline = rand_delay_start(S); t->sub = 1; }
            if (!sleep_poll(S, slot, t->deadline)) return 0;
            {
                if (t->conn >= 0) { int id = t->conn; t->conn = -1; conn_drop_handles(S, id, t->side, t->killed); t = &S->tasks.p[slot]; }
                uint64_t lat; int ds; unsigned lb;
                const addr_t dial = ipvs_rewrite(S, addr_of_sock(S, in->b & 0xff));    /* channel() is built from the rewritten dst */
                const int sent = try_send(S, w->socks[in->a].node, dial, &lat, &ds, &lb);
                if (sent < 0) return 1;
                if (!sent) {
                    t->val = MADSIM_VAL_REFUSED;           /* io::ErrorKind::ConnectionRefused */
                } else {
                    size_t id = 0;
                    while (id < S->conns.n && S->conns.p[id].alive) id++;
                    if (id >= MADSIM_MAX_CONNS) model_event(S, MADSIM_ORACLE_ME_CONNS);     /* the model's ceiling: a 128th live connection */
                    if (id == S->conns.n) { conn_t z; memset(&z, 0, sizeof z); vec_push(S->conns, z); }
                    conn_t* c = &S->conns.p[id];
                    c->alive = 1; c->c_node = w->socks[in->a].node; c->s_node = w->socks[ds].node;
                    c->dst_kind = dial.kind; c->dst_ipnode = dial.node; c->dst_port = dial.port;
                    c->src_kind = lb ? MADSIM_ADDR_LOOPBACK : MADSIM_ADDR_IP; c->src_port = S->socks[in->a].port;   /* src = (ip, port) :355 */
                    for (int d = 0; d < 2; d++) { c->d[d].tx_alive = c->d[d].rx_alive = 1; c->d[d].q.n = 0; c->d[d].rx_task = -1; }
                    c->guard_sock[0] = in->a; c->guard_sock[1] = -1;
                    if (S->socks[in->a].guards >= MADSIM_MAX_SOCKET_GUARDS) model_event(S, MADSIM_ORACLE_ME_GUARDS);
                    S->socks[in->a].guards++;              /* Sender { _guard: self.guard.clone(), .. }, Receiver { .. } (endpoint.rs:181-190) */
                    t->conn = (int8_t)id; t->side = 0; t->val = 0;
                    if (S->conns.n > S->st.max_conns) S->st.max_conns = (uint32_t)S->conns.n;
                    sock_t* k = &S->socks[ds];             /* socket.new_connection -> `let _ = conn_tx.try_send(..)` (endpoint.rs:320-328) */
                    if (!k->ep_alive) {
                        conn_drop_handles(S, (int)id, 1, 0);   /* the listener's Endpoint is gone (its address is held by connections
                                                                  it accepted): the channel is closed, (tx2, rx1) are dropped here */
                    } else {
                        /* conn_tx is an unbounded channel (endpoint.rs:307); the workload model holds eight connections waiting for
                         * accept1 per Endpoint (include/madsim_hip.h): a ninth leaves the model here (MADSIM_UNSUPPORTED, both sides) */
                        if (k->acceptq.n >= 8 && model_event(S, MADSIM_ORACLE_ME_ACCEPTQ)) return 1;
                        vec_push(k->acceptq, (uint8_t)id);
                        if (k->acc_task >= 0) { int32_t a = k->acc_task; k->acc_task = -1; wake(S, (uint16_t)a, k->acc_gen); }
                    }
                }
            }
            t = &S->tasks.p[slot]; t->sub = 0; t->pc++;
            break;
        case MS_OP_ACCEPT: {                               /* Endpoint::accept1 (endpoint.rs:197-211) */
            if (t->sub == 0) { t->deadline = rand_delay_start(S); t->sub = 1; }
            if (t->sub == 1) { if (!sleep_poll(S, slot, t->deadline)) return 0; t->sub = 2; }
            sock_t* k = &S->socks[in->a];
            if (k->acceptq.n == 0) { k->acc_task = slot; k->acc_gen = t->gen; return 0; }   /* conn_rx.recv() pending */
            /* `(tx, rx, _) = ep.accept1().await` over a pair already in hand: the right-hand side is evaluated first — the connection
             * leaves the queue — and the old pair drops on assignment.  (The other order also mis-sized the memmove when the drop
             * emptied this very queue: an Endpoint that was closed and is kept bound by that old pair alone.) */
            const int8_t taken = (int8_t)k->acceptq.p[0];
            memmove(k->acceptq.p, k->acceptq.p + 1, --k->acceptq.n);
            if (t->conn >= 0) { int id = t->conn; t->conn = -1; conn_drop_handles(S, id, t->side, t->killed); t = &S->tasks.p[slot]; k = &S->socks[in->a]; }
            t->conn = taken;
            if (k->guards >= MADSIM_MAX_SOCKET_GUARDS) model_event(S, MADSIM_ORACLE_ME_GUARDS);
            S->conns.p[t->conn].guard_sock[1] = in->a; k->guards++;   /* Sender / Receiver { _guard: self.guard.clone() } (endpoint.rs:203-210) */
            t->side = 1; t->sub = 0; t->pc++;
            break;
        }
        case MS_OP_CSEND: {                                /* Sender::send -> PayloadSender::send (net/mod.rs:417-421) */
            if (t->conn < 0) { t->val = MADSIM_VAL_RESET; t->pc++; break; }
            conn_t* c = &S->conns.p[t->conn];
            cdir_t* d = &c->d[t->side];
            cmsg_t m; m.val = in->imm; m.arrive = 0;
            const int link = chan_test_link(S, c, t->side, &m.arrive);             /* draws happen before the closed check */
            if (link < 0) return 1;                        /* `.ip.unwrap()` inside try_send (network.rs:309) */
            m.has_arrive = (uint8_t)link;
            if (!d->rx_alive) { t->val = MADSIM_VAL_RESET; t->pc++; break; }        /* ConnectionReset */
            if (d->q.n >= MADSIM_MAX_CHAN_QUEUE) model_event(S, MADSIM_ORACLE_ME_CHAN_QUEUE);   /* the model's ceiling: a 16th queued payload */
            vec_push(d->q, m);
            if (d->q.n > S->st.max_cq) S->st.max_cq = (uint32_t)d->q.n;
            if (d->rx_task >= 0) { int32_t r = d->rx_task; d->rx_task = -1; wake(S, (uint16_t)r, d->rx_gen); }
            t = &S->tasks.p[slot]; t->pc++;
            break;
        }
        case MS_OP_CRECV: {                                /* Receiver::recv -> the stream of channel() (net/mod.rs:385-402) */
            if (t->conn < 0) { t->val = MADSIM_VAL_RESET; t->pc++; break; }
            conn_t* c = &S->conns.p[t->conn];
            cdir_t* d = &c->d[1 - t->side];
            if (t->sub == 0) {                             /* rx.recv().await */
                if (d->q.n == 0) {
                    if (!d->tx_alive) { t->val = MADSIM_VAL_RESET; t->pc++; break; }
                    d->rx_task = slot; d->rx_gen = t->gen;
                    return 0;
                }
                cmsg_t m = d->q.p[0];
                memmove(d->q.p, d->q.p + 1, (--d->q.n) * sizeof(cmsg_t));
                t->cval = m.val; t->chas = m.has_arrive; t->carrive = m.arrive; t->backoff_ms = 1;
                t->sub = 1;
            }
            for (;;) {
                if (t->sub == 1) {
                    if (t->chas) { t->deadline = sleep_deadline(S, t->carrive); t->sub = 3; }          /* sleep_until(arrive_time) */
                    else { t->deadline = sleep_deadline(S, S->clock + (uint64_t)t->backoff_ms * NS_PER_MS); t->sub = 2; }  /* sleep(backoff) */
                }
                if (!sleep_poll(S, slot, t->deadline)) return 0;
                if (t->sub == 3) break;
                t->backoff_ms = t->backoff_ms * 2 > 10000 ? 10000 : t->backoff_ms * 2;               /* min(backoff * 2, 10 s) */
                const int link = chan_test_link(S, c, 1 - t->side, &t->carrive);                     /* retry */
                if (link < 0) return 1;
                t->chas = (uint8_t)link;
                t->sub = 1;
            }
            t->val = t->cval; t->sub = 0; t->pc++;
            break;
        }
        case MS_OP_CCLOSE:
            if (t->conn >= 0) { int id = t->conn; t->conn = -1; conn_drop_handles(S, id, t->side, t->killed); t = &S->tasks.p[slot]; }
            t->pc++;
            break;
        case MS_OP_ASSERT_VAL:
            if (t->val != in->imm) return 1;
            t->pc++;
            break;
        case MS_OP_JEQ:                                    /* if val == imm { goto b } */
            t->pc = (t->val == in->imm) ? in->b : t->pc + 1;
            break;
        case MS_OP_SLEEP_RAND:                             /* sleep(thread_rng().gen_range(lo..hi)).await:
                                                              tonic-example/tests/test.rs:199; UniformDuration on the
                                                              GlobalRng itself => one with() per attempt (A.3/A.4) */
            if (t->sub == 0) {
                int mode; uint64_t low, range, zone;
                oracle_uniform_duration_params((uint64_t)in->a * 50 * NS_PER_MS, insn_dur(in), &mode, &low, &range, &zone);
                uint64_t d = sample_duration(S, mode, low, range, zone, 1);
                t->deadline = sleep_deadline(S, S->clock + d);
                t->sub = 1;
            }
            if (!sleep_poll(S, slot, t->deadline)) return 0;
            t->sub = 0; t->pc++;
            break;
        case MS_OP_RECV_TIMEOUT: {                         /* timeout(d, ep.recv_from(tag)) — time/mod.rs:128-140:
                                                              select_biased! { fut, sleep }: the inner future is polled
                                                              first, then the timeout's Sleep, which registers ANOTHER
                                                              timer on every not-elapsed poll (time/sleep.rs:51-53) */
            sock_t* k = &S->socks[in->a];
            uint8_t tag = (uint8_t)(in->b >> 8);
            if (t->sub == 0) {
                t->deadline2 = sleep_deadline(S, S->clock + (uint64_t)(in->b & 0xff) * NS_PER_S + in->imm);
                t->rxseq++; t->inbox_full = 0;
                size_t idx = 0;
                while (idx < k->msgs.n && k->msgs.p[idx].tag != tag) idx++;
                if (idx < k->msgs.n) {
                    msg_t m = k->msgs.p[idx];
                    k->msgs.p[idx] = k->msgs.p[--k->msgs.n];
                    t->inbox_full = 1; t->val = m.val; t->from = m.from; t->inbox_aux = m.aux;
                } else {
                    reg_t r = { tag, slot, t->gen, t->rxseq, (uint8_t)tag };
                    reg_model_limits(S, k, &r);
                    vec_push(k->registered, r);
                    if (k->registered.n > S->st.max_regs) S->st.max_regs = (uint32_t)k->registered.n;
                }
                t->sub = 1;
            }
            int ready = 0;
            if (t->sub == 1 && t->inbox_full) {            /* oneshot ready -> rand_delay (endpoint.rs:145) */
                t->inbox_full = 0;
                if (tag >= MADSIM_TAG_RPC_FIRST) t->aux = t->inbox_aux;
                t->deadline = rand_delay_start(S); t->sub = 2;
            }
            if (t->sub == 2) ready = sleep_poll(S, slot, t->deadline);
            if (ready) { t->sub = 0; t->pc++; break; }     /* Ok((len, from)) */
            if (S->clock >= t->deadline2) {                /* Err(Elapsed): the recv future is dropped */
                t->rxseq++; t->inbox_full = 0;             /* oneshot::Receiver gone; a message taken in sub 2 is lost */
                t->val = MADSIM_VAL_TIMEOUT;
                t->sub = 0; t->pc++;
                break;
            }
            { event_t e; memset(&e, 0, sizeof e);          /* Sleep::poll of the timeout: a NEW timer every time */
              e.deadline = t->deadline2; e.kind = EV_WAKE; e.slot = slot; e.gen = t->gen; timer_add(S, e); }
            return 0;
        }
        case MS_OP_RPC_CALL: {                             /* Endpoint::call / call_timeout (rpc.rs:96-131) */
            sock_t* k = &S->socks[in->a];
            const uint32_t timeout_ms = in->imm >> 8;
            const unsigned dst = in->b & 0xff;
            int ready = 0;
            if (t->sub == 0) {
                /* timeout(d, self.call(..)) builds its Sleep before the call future is first polled (time/mod.rs:128-133) */
                if (timeout_ms) t->deadline2 = sleep_deadline(S, S->clock + (uint64_t)timeout_ms * NS_PER_MS);
                t->rsp_tag = rng_next(S); rng_log(S);      /* random::<u64>() on the GlobalRng: one with() (rand.rs:146-148) */
                t->deadline = rand_delay_start(S);         /* send_to_raw -> NetSim::send: rand_delay first (net/mod.rs:306) */
                t->sub = 1;
            }
            if (t->sub == 1 && sleep_poll(S, slot, t->deadline)) {
                uint64_t lat; int ds;
                /* hooks_req.get(&node): `if !hook(&msg) { return Ok(()) }` before try_send (net/mod.rs:307-311) */
                const node_t* sn = &S->nodes[w->socks[in->a].node];
                const int hooked = sn->hreq_valid && sn->hreq_tag == (uint8_t)(in->b >> 8) && (sn->hreq_all || sn->hreq_code == (uint8_t)in->imm);
                unsigned lb = 0;
                const int sent = hooked ? 0 : try_send(S, w->socks[in->a].node, ipvs_rewrite(S, addr_of_sock(S, dst)), &lat, &ds, &lb);
                if (sent < 0) return 1;
                if (sent) {
                    event_t e; memset(&e, 0, sizeof e);
                    e.deadline = S->clock + lat; e.kind = EV_DELIVER; e.sock = (uint8_t)ds;
                    e.sockgen = S->socks[ds].gen; e.from = mk_from(S, in->a, lb); e.tag = (uint8_t)(in->b >> 8);
                    e.val = in->imm & 0xff; e.aux = t->rsp_tag;            /* Box::new((rsp_tag, request, data)) */
                    timer_add(S, e);
                    t = &S->tasks.p[slot];
                }
                /* recv_from_raw(rsp_tag): Mailbox::recv (endpoint.rs:353-362) */
                t->rxseq++; t->inbox_full = 0;
                size_t idx = 0;
                while (idx < k->msgs.n && k->msgs.p[idx].tag != t->rsp_tag) idx++;
                if (idx < k->msgs.n) {
                    msg_t m = k->msgs.p[idx];
                    k->msgs.p[idx] = k->msgs.p[--k->msgs.n];
                    t->inbox_full = 1; t->val = m.val; t->from = m.from; t->inbox_aux = m.aux;
                } else {
                    reg_t r = { t->rsp_tag, slot, t->gen, t->rxseq, 0xff };
                    reg_model_limits(S, k, &r);
                    vec_push(k->registered, r);
                    if (k->registered.n > S->st.max_regs) S->st.max_regs = (uint32_t)k->registered.n;
                }
                t->sub = 2;
            }
            if (t->sub == 2 && t->inbox_full) {            /* oneshot ready -> rand_delay (endpoint.rs:145) */
                t->inbox_full = 0;
                t->deadline = rand_delay_start(S); t->sub = 3;
            }
            if (t->sub == 3) ready = sleep_poll(S, slot, t->deadline);
            if (ready) {
                if (!addr_eq(addr_of_from(S, t->from), addr_of_sock(S, dst))) return 1;   /* assert_eq!(from, dst) rpc.rs:126 */
                t->sub = 0; t->pc++;
                break;
            }
            if (!timeout_ms) return 0;
            if (S->clock >= t->deadline2) {                /* Err(Elapsed) -> io::ErrorKind::TimedOut: the call future is dropped */
                if (t->sub >= 2) { t->rxseq++; t->inbox_full = 0; }     /* its oneshot::Receiver with it */
                t->val = MADSIM_VAL_TIMEOUT;
                t->sub = 0; t->pc++;
                break;
            }
            { event_t e; memset(&e, 0, sizeof e);          /* Sleep::poll of the timeout: a NEW timer every time */
              e.deadline = t->deadline2; e.kind = EV_WAKE; e.slot = slot; e.gen = t->gen; timer_add(S, e); }
            return 0;
        }
        case MS_OP_CLOSE: {
            sock_t* k = &S->socks[in->a];
            if (k->ep_alive && k->owner_slot == slot && k->owner_gen == t->gen) { endpoint_drop(S, k, t->killed); t = &S->tasks.p[slot]; }
            t->pc++;
            break;
        }
        case MS_OP_CLOG_NODE:                              /* network.rs:162-171 */
            if (in->b & 1) S->clog_in |= 1ull << in->a;
            if (in->b & 2) S->clog_out |= 1ull << in->a;
            t->pc++;
            break;
        case MS_OP_UNCLOG_NODE:                            /* network.rs:173-182 */
            if (in->b & 1) S->clog_in &= ~(1ull << in->a);
            if (in->b & 2) S->clog_out &= ~(1ull << in->a);
            t->pc++;
            break;
        case MS_OP_CLOG_LINK:
            S->clog_link[in->a] |= 1ull << in->b; t->pc++;
            break;
        case MS_OP_UNCLOG_LINK:
            S->clog_link[in->a] &= ~(1ull << in->b); t->pc++;
            break;
        case MS_OP_SET_LOSS: {                             /* net/mod.rs:138-141 */
            double p = S->cfg->loss_table[in->a & 3];
            S->loss_always = p == 1.0;
            S->loss_pint = S->loss_always ? 0 : (uint64_t)(p * 18446744073709551616.0);
            t->pc++;
            break;
        }
        case MS_OP_SET_LATENCY: {                          /* NetSim::update_config(|c| c.send_latency = lo..hi): net/mod.rs:138-141 ->
                                                              Network::update_config (network.rs:129); test_link samples
                                                              `self.config.send_latency.clone()` on every call (:267) */
            const uint32_t k = in->a & 3;
            oracle_uniform_duration_params(S->cfg->lat_table_lo_ns[k], S->cfg->lat_table_hi_ns[k], &S->lat_mode, &S->lat_low, &S->lat_range, &S->lat_zone);
            t->pc++;
            break;
        }
        case MS_OP_RANDOM: {                               /* one with() on the GlobalRng's RngCore impl (rand.rs:142-158) */
            uint64_t v = rng_next(S); rng_log(S);
            /* a=0: gen::<u32>() = next_u32 = upper half [DEP A.1]; a=1: fill_bytes of 1 byte = first LE byte of next_u32
             * [DEP rand_core 0.6 fill_bytes_via_next: a tail of <= 4 bytes takes next_u32] */
            t->val = in->a == 0 ? (uint32_t)(v >> 32) : (uint32_t)((v >> 32) & 0xff);
            t->pc++;
            break;
        }
        case MS_OP_TRACE_TIME: {                           /* SystemTime::now() / Instant::now() (time/system_time.rs) */
            uint64_t v = in->a == 0 ? S->base_time_ns + S->clock : in->a == 1 ? S->clock : t->val;
            obs_record(v); S->obs_hash = (S->obs_hash ^ v) * FNV_PRIME; t->pc++;
            break;
        }
        case MS_OP_RAND_BOOL: {                            /* thread_rng().gen_bool(p): Bernoulli on the GlobalRng [DEP A.4] */
            double p = S->cfg->loss_table[in->a & 3];
            int always = p == 1.0;
            t->val = (uint32_t)gen_bool_pint(S, always ? 0 : (uint64_t)(p * 18446744073709551616.0), always);
            t->pc++;
            break;
        }
        default:
            return 1;                                      /* unsupported op in this oracle build */
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Timer::expire                                                        [DEP A.5], time/mod.rs:54,105
 * ---------------------------------------------------------------------------------------------- */
static void timer_expire(sim_t* S, uint64_t now) {
    while (S->heap.n > 0 && S->heap.p[0].deadline <= now) {
        event_t e = timer_pop(S);
        S->steps++;
        switch (e.kind) {
        case EV_WAKE: wake(S, e.slot, e.gen); break;       /* time/sleep.rs:52 waker.wake() */
        case EV_DELIVER:                                   /* net/mod.rs:323-330 */
            if (e.is_rsp && e.hook_valid && (e.hook_all || e.hook_code == (uint8_t)e.val)) break;   /* !hook(&msg): return */
            mailbox_deliver(S, &e); break;
        case EV_RESTART: node_restart(S, e.node); break;   /* task/mod.rs:313 */
        default: break;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Executor::run_all_ready                                              task/mod.rs:263-323
 * ---------------------------------------------------------------------------------------------- */
static void run_all_ready(sim_t* S, uint32_t max_steps) {
    while (S->ready.n > 0 && !S->panic && S->steps < max_steps) {
        /* try_recv_random (utils/mpsc.rs:73-83): idx drawn even when len == 1 */
        size_t idx = (size_t)gen_range_u64(S, 0, S->ready.n);
        uint16_t slot = S->ready.p[idx];
        S->ready.p[idx] = S->ready.p[--S->ready.n];        /* swap_remove */
        task_t* t = &S->tasks.p[slot];
        S->steps++;
        if (t->cancelled || t->killed) {                   /* :269-273 drop(runnable) */
            task_finish(S, slot, H_CANCELLED);
        } else if (S->nodes[t->node].paused) {             /* :274-277 */
            vec_push(S->nodes[t->node].paused_list, slot);
            S->steps--;                                    /* no poll, no time advance */
            continue;
        } else {
            t->scheduled = 0; t->running = 1;              /* async-task run(): clear SCHEDULED, set RUNNING */
            S->panic_code = MADSIM_PANIC_CODE_OTHER;      /* failed asserts / unwraps: a message no pattern names */
            int panicked = poll_task(S, slot);
            t = &S->tasks.p[slot];
            if (panicked) {                                /* :289-317 */
                unsigned node = t->node;
                const madsim_node_t* nb = &S->w->nodes[node];
                /* restart_on_panic || restart_on_panic_matching.iter().any(|s| error_msg.contains(s)) (:297-300) */
                int restart = (nb->flags & MADSIM_NODE_RESTART_ON_PANIC) != 0;
                if (nb->flags & MADSIM_NODE_RESTART_MATCHING) {
                    if (S->w->panic_match)                 /* the host evaluated `error_msg.contains(pattern)` for every message code */
                        restart |= (S->w->panic_match[node * 8 + (S->panic_code >> 5)] >> (S->panic_code & 31)) & 1;
                    else for (unsigned k = 0; k < nb->n_match && k < 2; k++) restart |= nb->match[k] == S->panic_code;
                }
                if (!restart || S->unsupported) {
                    S->panic = 1;
                    return;                                /* resume_unwind: block_on unwinds */
                }
                /* async-task's panic guard: the future is dropped, the task closed, the awaiter notified */
                task_finish(S, slot, H_CANCELLED);
                /* delay = gen_range(1 s..10 s) inside ONE with() (:302-304): UniformDuration Medium path */
                int mode; uint64_t low, range, zone;
                oracle_uniform_duration_params(1 * NS_PER_S, 10 * NS_PER_S, &mode, &low, &range, &zone);
                uint64_t delay = sample_duration(S, mode, low, range, zone, 0);
                node_kill(S, node);                        /* self.kill(node_id) :309 */
                event_t e; memset(&e, 0, sizeof e);
                e.deadline = S->clock + delay; e.kind = EV_RESTART; e.node = (uint8_t)node;
                timer_add(S, e);                           /* add_timer(delay, restart(node)) :311-313 */
                t = &S->tasks.p[slot];
            }
            if (t->alive) {
                t->running = 0;
                if (t->scheduled) ready_push(S, slot);     /* woken while running: re-queue after the poll */
            }
        }
        /* :319-321 advance time 50..100 ns, then Timer::expire (time/mod.rs:103-106) */
        uint64_t dur = gen_range_u64(S, 50, 100);
        S->clock += dur;
        timer_expire(S, S->clock);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Runtime::with_seed_and_config + block_on                             runtime/mod.rs:53-69,127-130
 * ---------------------------------------------------------------------------------------------- */
static int validate(const madsim_workload_t* w, const madsim_config_t* cfg) {
    if (!w || !w->insns || !w->progs || w->n_progs == 0 || w->n_progs > 255) return -1;
    if (w->n_nodes > 62 || w->n_socks > 63) return -1;
    if (w->n_socks && !w->socks) return -1;
    for (uint32_t i = 0; i < w->n_socks; i++) if (w->socks[i].kind > MADSIM_ADDR_VIRTUAL) return -1;
    if (w->n_services > MADSIM_MAX_SERVICES || (w->n_services && !w->services)) return -1;
    for (uint32_t k = 0; k < w->n_services; k++) {
        const uint32_t ns = w->services[k].n_servers;
        if (w->services[k].vaddr >= w->n_socks || (ns > 6 && ns != MADSIM_SERVICE_ABSENT)) return -1;
        for (uint32_t j = 0; j < (ns & 7u); j++) if (w->services[k].servers[j] >= w->n_socks) return -1;
    }
    for (uint32_t i = 0; i < w->n_insns; i++) {           /* an ephemeral Endpoint has no address a peer could name */
        const madsim_insn_t* in = &w->insns[i];
        if (in->op == MS_OP_BIND && in->a < w->n_socks && w->socks[in->a].kind == MADSIM_ADDR_VIRTUAL) return -1;   /* a destination only */
        if (in->op == MS_OP_IPVS && (in->a > MADSIM_IPVS_DEL_SERVER || in->b >= w->n_services ||
                                     (in->a >= MADSIM_IPVS_ADD_SERVER && (in->imm >= w->n_socks || w->socks[in->imm].port == 0)))) return -1;
        if ((in->op == MS_OP_SEND || in->op == MS_OP_CONNECT || in->op == MS_OP_RPC_CALL) &&
            (uint32_t)(in->b & 0xff) < w->n_socks && w->socks[in->b & 0xff].port == 0) return -1;
    }
    if (w->n_insns == 0) return -1;
    {                                                     /* no body may run off the end of the table (the kernel does not check) */
        const uint8_t last = w->insns[w->n_insns - 1].op;
        if (last != MS_OP_DONE && last != MS_OP_JMP && last != MS_OP_PANIC) return -1;
        for (uint32_t i = 0; i < w->n_insns; i++)
            if ((w->insns[i].op == MS_OP_DJNZ || w->insns[i].op == MS_OP_JMP || w->insns[i].op == MS_OP_JEQ) && w->insns[i].b >= w->n_insns) return -1;
        for (uint32_t p = 0; p < w->n_progs; p++) if (w->progs[p].entry >= w->n_insns) return -1;
    }
    for (uint32_t p = 0; p < w->n_progs; p++) {           /* a guard's Drop spawns the next program, on the same node */
        if (!(w->progs[p].flags & MADSIM_PROG_DROP_SPAWN)) continue;
        if (p + 1 >= w->n_progs || w->progs[p + 1].node != w->progs[p].node) return -1;
        for (uint32_t i = 0; i < w->n_insns; i++) if (w->insns[i].op == MS_OP_PAUSE) return -1;
    }
    {   /* reset_node's socket drop order (network.rs:142-147: a HashMap under the seed's SipHash keys) is not restated: workloads
         * where it could be observed — a resettable node with two listening Endpoints — are refused, here as in the library */
        uint64_t resettable = 0;
        for (uint32_t i = 0; i < w->n_insns; i++) if (w->insns[i].op == MS_OP_KILL || w->insns[i].op == MS_OP_RESTART) resettable |= 1ull << (w->insns[i].a & 63);
        for (uint32_t n = 0; n <= w->n_nodes && w->nodes; n++) if (w->nodes[n].flags & (MADSIM_NODE_RESTART_ON_PANIC | MADSIM_NODE_RESTART_MATCHING)) resettable |= 1ull << n;
        for (uint32_t p = 0; p < w->n_progs; p++) if (w->progs[p].flags & MADSIM_PROG_INIT) resettable |= 1ull << (w->progs[p].node & 63);
        for (uint32_t n = 1; n <= w->n_nodes && n < 64; n++) {
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < w->n_insns; i++) {
                if (w->insns[i].op != MS_OP_ACCEPT || w->insns[i].a >= w->n_socks || w->socks[w->insns[i].a].node != n) continue;
                int seen = 0;
                for (uint32_t j = 0; j < i; j++) seen |= w->insns[j].op == MS_OP_ACCEPT && w->insns[j].a == w->insns[i].a;
                cnt += !seen;
            }
            if (((resettable >> n) & 1) && cnt >= 2) return -1;
        }
    }
    if (cfg->lat_lo_ns >= cfg->lat_hi_ns) return -1;      /* "cannot sample empty range" */
    if (!(cfg->packet_loss_rate >= 0.0 && cfg->packet_loss_rate <= 1.0)) return -1;
    if (cfg->n_lat_table > 4) return -1;
    for (uint32_t k = 0; k < cfg->n_lat_table; k++) if (cfg->lat_table_lo_ns[k] >= cfg->lat_table_hi_ns[k]) return -1;   /* an empty range would panic at the next send */
    for (uint32_t i = 0; i < w->n_insns; i++) if (w->insns[i].op == MS_OP_SET_LATENCY && w->insns[i].a >= cfg->n_lat_table) return -1;
    return 0;
}

static void run_one(const madsim_workload_t* w, const madsim_config_t* cfg, const madsim_limits_t* lim,
                    uint64_t seed, madsim_result_t* out, uint8_t* log, uint64_t log_cap,
                    uint64_t* log_len, madsim_oracle_stats_t* stats, int model_limits, uint32_t* events) {
    sim_t S; memset(&S, 0, sizeof S);
    S.w = w; S.cfg = cfg; S.model_limits = model_limits;
    S.handles = calloc(w->n_progs, sizeof *S.handles);
    S.nodes = calloc(w->n_nodes + 1, sizeof *S.nodes);
    S.socks = calloc(w->n_socks ? w->n_socks : 1, sizeof *S.socks);
    S.clog_link = calloc(w->n_nodes + 1, sizeof *S.clog_link);
    for (uint32_t i = 0; i < w->n_socks; i++) {
        S.socks[i].acc_task = -1;
        S.socks[i].port = w->socks[i].port;               /* 0: ephemeral, set by its bind (never a destination: validate) */
    }
    for (uint32_t k = 0; k < w->n_services; k++) {        /* the table = the ipvs calls made before the first task runs */
        const madsim_service_t* sv = &w->services[k];
        S.ipvs_present[k] = !(sv->n_servers & MADSIM_SERVICE_ABSENT);
        S.ipvs_n[k] = sv->n_servers & 7u;
        for (uint32_t j = 0; j < S.ipvs_n[k]; j++) S.ipvs_srv[k][j] = sv->servers[j];
    }
    S.trace_hash = FNV_OFFSET; S.obs_hash = FNV_OFFSET;
    S.log = log; S.log_cap = log_cap;
    S.no_log = lim && lim->no_trace_hash && !log;          /* a trace request always logs, like madsim_hip_trace_seed */
    S.buggify = cfg->buggify != 0;
    S.loss_always = cfg->packet_loss_rate == 1.0;
    S.loss_pint = S.loss_always ? 0 : (uint64_t)(cfg->packet_loss_rate * 18446744073709551616.0);
    oracle_uniform_duration_params(cfg->lat_lo_ns, cfg->lat_hi_ns, &S.lat_mode, &S.lat_low, &S.lat_range, &S.lat_zone);
    uint32_t max_steps = lim && lim->max_steps ? lim->max_steps : (1u << 24);
    uint64_t time_limit = lim ? lim->time_limit_ns : 0;

    /* GlobalRng::new_with_seed (rand.rs:42-61) */
    oracle_seed_from_u64(seed, S.rng.s);
    /* TimeRuntime::new (time/mod.rs:26-38): base_time draw.  Logging is enabled only after
     * construction (runtime/mod.rs:185-186), so this draw is not in the determinism log. */
    {
        uint64_t h = S.trace_hash, n = S.log_len;
        S.base_time_ns = (60ull * 60 * 24 * 365 * (2022 - 1970) + gen_range_u64(&S, 0, 60ull * 60 * 24 * 365)) * NS_PER_S;
        S.trace_hash = h; S.log_len = n;
    }
    /* Tasks spawned BEFORE block_on (the `runtime.create_node()..build(); node.spawn(..);
     * runtime.block_on(..)` shape of task/mod.rs:859-897): they enter the ready Vec ahead of main. */
    for (uint32_t p = 1; p < w->n_progs; p++)
        if (w->progs[p].flags & MADSIM_PROG_PRE) spawn_task(&S, p, !(w->progs[p].flags & MADSIM_PROG_INIT));

    /* Executor::block_on (task/mod.rs:220-260) */
    S.main_slot = spawn_task(&S, 0, 1);
    int verdict = MADSIM_PASS;
    for (;;) {
        run_all_ready(&S, max_steps);
        if (S.panic) { verdict = MADSIM_PANIC; break; }
        if (S.steps >= max_steps) { verdict = MADSIM_STEP_LIMIT; break; }
        if (S.handles[0].state != H_RUNNING) break;        /* task.is_finished() :241-243 */
        /* TimeRuntime::advance_to_next_event (time/mod.rs:45-60) */
        if (S.heap.n == 0) { verdict = MADSIM_DEADLOCK; break; }      /* :250 */
        uint64_t t = S.heap.p[0].deadline + 50;
        timer_expire(&S, t);                               /* callbacks run before the clock is set */
        S.clock = t;
        if (time_limit && !(S.clock < time_limit)) { verdict = MADSIM_TIME_LIMIT; break; }  /* :253-258 */
    }
    out->verdict = (uint32_t)verdict; out->steps = S.steps; out->clock_ns = S.clock;
    out->msg_count = S.msg_count; out->rng_calls = S.rng_calls; out->trace_hash = S.no_log ? 0 : S.trace_hash;
    out->obs_hash = S.obs_hash;
    if (S.unsupported) { memset(out, 0, sizeof *out); out->verdict = MADSIM_UNSUPPORTED; }   /* the verdict is the whole answer */
    if (log_len) *log_len = S.log_len;
    if (events) *events = S.model_events;
    if (stats) {
        if (S.st.max_heap > stats->max_heap) stats->max_heap = S.st.max_heap;
        if (S.st.max_ready > stats->max_ready) stats->max_ready = S.st.max_ready;
        if (S.st.max_tasks > stats->max_tasks) stats->max_tasks = S.st.max_tasks;
        if (S.st.max_msgs > stats->max_msgs) stats->max_msgs = S.st.max_msgs;
        if (S.st.max_regs > stats->max_regs) stats->max_regs = S.st.max_regs;
        if (S.st.max_conns > stats->max_conns) stats->max_conns = S.st.max_conns;
        if (S.st.max_cq > stats->max_cq) stats->max_cq = S.st.max_cq;
    }
    for (uint32_t i = 0; i < w->n_socks; i++) { vec_free(S.socks[i].registered); vec_free(S.socks[i].msgs); vec_free(S.socks[i].acceptq); }
    for (size_t i = 0; i < S.conns.n; i++) { vec_free(S.conns.p[i].d[0].q); vec_free(S.conns.p[i].d[1].q); }
    vec_free(S.conns);
    for (uint32_t i = 0; i <= w->n_nodes; i++) { vec_free(S.nodes[i].paused_list); vec_free(S.nodes[i].tasks); }
    vec_free(S.heap); vec_free(S.ready); vec_free(S.tasks);
    free(S.handles); free(S.nodes); free(S.socks); free(S.clog_link);
}

/* ------------------------------------------------------------------------------------------------
 * public API (mirrors madsim_hip_run_batch)
 * ---------------------------------------------------------------------------------------------- */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

int madsim_oracle_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0,
                            uint64_t count, const madsim_limits_t* lim, madsim_result_t* out,
                            madsim_summary_t* summary, madsim_oracle_stats_t* stats) {
    if (!cfg || validate(w, cfg)) return MADSIM_E_ARG;
    double t0 = now_s();
    uint64_t first = UINT64_MAX, nfail = 0, tsteps = 0, tclock = 0;
    for (uint64_t i = 0; i < count; i++) {
        madsim_result_t r;
        run_one(w, cfg, lim, seed0 + i, &r, NULL, 0, NULL, stats, 1, NULL);
        if (out) out[i] = r;
        if (r.verdict != MADSIM_PASS) { nfail++; if (seed0 + i < first) first = seed0 + i; }
        tsteps += r.steps; tclock += r.clock_ns;
    }
    if (summary) {
        summary->first_failing_seed = first; summary->n_failed = nfail; summary->total_steps = tsteps;
        summary->total_clock_ns = tclock; summary->kernel_ms = 0.0; summary->wall_s = now_s() - t0;
    }
    return 0;
}

int64_t madsim_oracle_trace_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                                 const madsim_limits_t* lim, uint8_t* log, uint64_t cap,
                                 madsim_result_t* out) {
    if (!cfg || validate(w, cfg)) return MADSIM_E_ARG;
    madsim_result_t r; uint64_t n = 0;
    run_one(w, cfg, lim, seed, &r, log, cap, &n, NULL, 1, NULL);
    if (out) *out = r;
    return (int64_t)n;
}

/* The restatement WITHOUT the workload model's ceilings (see model_event): results as the reference's unbounded containers give
 * them, plus per seed the mask of model events it met.  A seed with events[i] == 0 never touched a ceiling: its result is what
 * madsim_oracle_run_batch reports; for any other seed the device runner's answer is the verdict MADSIM_UNSUPPORTED. */
int madsim_oracle_run_batch_pure(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                 const madsim_limits_t* lim, madsim_result_t* out, uint32_t* events) {
    if (!cfg || validate(w, cfg)) return MADSIM_E_ARG;
    for (uint64_t i = 0; i < count; i++) {
        madsim_result_t r; uint32_t ev = 0;
        run_one(w, cfg, lim, seed0 + i, &r, NULL, 0, NULL, NULL, 0, &ev);
        if (out) out[i] = r;
        if (events) events[i] = ev;
    }
    return 0;
}

/* [DEP rand 0.8 gen_range on u64] exposed for the known-answer tests. */
uint64_t madsim_oracle_gen_range(uint64_t s[4], uint64_t lo, uint64_t hi, uint64_t* ncalls) {
    uint64_t range = hi - lo, zone = (range << __builtin_clzll(range)) - 1, n = 0, res;
    for (;;) {
        uint64_t v = oracle_xoshiro_next(s); n++;
        unsigned __int128 m = (unsigned __int128)v * range;
        if ((uint64_t)m <= zone) { res = lo + (uint64_t)(m >> 64); break; }
    }
    if (ncalls) *ncalls = n;
    return res;
}

/* The CPU twin SURVEY.md §8b asks for: the identical signature of madsim_hip_run_batch (include/madsim_hip.h), so a
 * host can swap one symbol for the other.  Test infrastructure like the rest of this file. */
int madsim_cpu_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                         const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary) {
    return madsim_oracle_run_batch(w, cfg, seed0, count, lim, out, summary, NULL);
}

/* One seed with its observed-value list (see obs_record).  Returns the number of observations (may exceed cap). */
int64_t madsim_oracle_observe_seed(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed,
                                   const madsim_limits_t* lim, uint64_t* obs, uint64_t cap, madsim_result_t* out) {
    if (!cfg || validate(w, cfg)) return MADSIM_E_ARG;
    madsim_result_t r;
    g_obs_buf = obs; g_obs_cap = cap; g_obs_len = 0;
    run_one(w, cfg, lim, seed, &r, NULL, 0, NULL, NULL, 1, NULL);
    g_obs_buf = NULL;
    if (out) *out = r;
    return (int64_t)g_obs_len;
}
