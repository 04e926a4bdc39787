// geometry.h — host-only (no HIP calls): workload validation and the LDS geometry the kernel runs with.
// Shared by madsim_hip.cpp and by the device-code emulation harness under tests/emu (debug aid only).
#ifndef MADSIM_GEOMETRY_H
#define MADSIM_GEOMETRY_H

#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "sim_kernel.h"

namespace madsim_geo {

using madsim_k::KParams;

// vgprs: VGPRs per lane of a kernel build as the loaded code object reports them (null / 0 = unknown: a static estimate)
// max_waves_per_simd: experiment hook (MADSIM_HIP_WAVES_PER_SIMD in the environment of the library): global-state builds size their LDS
// heap quota for at most that many waves per SIMD instead of what the build's registers admit (0 = no cap)
struct Device { int num_cus = 256; size_t lds_per_cu = 160 * 1024; int (*vgprs)(const madsim_k::VariantSel*) = nullptr; int max_waves_per_simd = 0; };

inline int fail(std::string* err, int code, const std::string& msg) { if (err) *err = msg; return code; }

struct Geo {
    KParams P;
    uint32_t lds_bytes, lds_per_seed, blocks_per_cu, grid, lanes_per_wave, waves_per_block;
};

inline bool uses_op(const madsim_workload_t* w, int op) {
    for (uint32_t i = 0; i < w->n_insns; i++) if (w->insns[i].op == op) return true;
    return false;
}

// UniformDuration::new(lo, hi) [DEP rand 0.8]: see SURVEY.md Appendix A.3
inline void uniform_duration_params(uint64_t lo, uint64_t hi, uint32_t* mode, uint64_t* low, uint64_t* range, uint64_t* zone) {
    const uint64_t S = 1000000000ull;
    uint64_t h = hi - 1;
    uint64_t lo_s = lo / S, lo_n = lo % S, hi_s = h / S, hi_n = h % S;
    if (hi_n < lo_n) { hi_s -= 1; hi_n += S; }
    if (lo_s == hi_s) {
        uint32_t r = (uint32_t)(hi_n - lo_n + 1);
        uint32_t reject = r ? (uint32_t)((0xffffffffu - r + 1u) % r) : 0;
        *mode = 0; *low = lo_s * S + lo_n; *range = r; *zone = 0xffffffffu - reject;
    } else {
        uint64_t r = h - lo + 1;
        uint64_t reject = r ? (UINT64_MAX - r + 1) % r : 0;
        *mode = 1; *low = lo; *range = r; *zone = UINT64_MAX - reject;
    }
}

inline void bernoulli(double p, uint64_t* p_int, uint32_t* always) {   // [DEP rand 0.8 Bernoulli::new]
    *always = p == 1.0;
    *p_int = *always ? 0 : (uint64_t)(p * 18446744073709551616.0);
}

// The device socket table.  A caller's entry is one word, node | kind << 8 | port << 16.  An entry with port 0 is an
// EPHEMERAL Endpoint (`Endpoint::bind("0.0.0.0:0")`): Network::bind gives it the lowest port from 1 up that no socket of
// the node holds for the same IP (network.rs:224-236).  On the device such an entry is a HANDLE, node | (kind | 0x80) << 8
// | base << 16 | K << 24, over K candidate entries base .. base+K-1 that this function appends for its (node, IP):
// ordinary entries with the ports 1 .. K, where K = the number of entries of that (node, IP) — at most K-1 other sockets
// can be bound when the handle binds, so a free candidate always exists, as it does among 65 535 ports — and twice that in
// workloads with connections: an address outlives its Endpoint while a Sender / Receiver made from it is alive
// (k_channel.h guard_release), so each entry may have one such predecessor still in the table.  Binding a
// handle binds the candidate with the lowest free port (every other op on the handle is redirected to the candidate
// bound last, k_state.h sock_resolve), so the socket a message comes from or goes to is always an ordinary entry with a
// fixed address and nothing else in the kernel knows about ephemeral ports.
inline std::vector<uint32_t> device_socks(const madsim_workload_t* w) {
    std::vector<uint32_t> t(w->n_socks);
    for (uint32_t i = 0; i < w->n_socks; i++) t[i] = (uint32_t)w->socks[i].node | ((uint32_t)w->socks[i].kind << 8) | ((uint32_t)w->socks[i].port << 16);
    for (uint32_t i = 0; i < w->n_socks; i++) {
        if (w->socks[i].port != 0 || (t[i] & 0x8000u)) continue;
        uint32_t K = 0;
        for (uint32_t j = 0; j < w->n_socks; j++) K += w->socks[j].node == w->socks[i].node && w->socks[j].kind == w->socks[i].kind;
        if (uses_op(w, MS_OP_CONNECT)) K *= 2;
        const uint32_t base = (uint32_t)t.size();
        if (base + K > 255) { t.resize(256); return t; }                         // far too many: validate() rejects it
        for (uint32_t p = 1; p <= K; p++) t.push_back((uint32_t)w->socks[i].node | ((uint32_t)w->socks[i].kind << 8) | (p << 16));
        for (uint32_t j = i; j < w->n_socks; j++)
            if (w->socks[j].port == 0 && w->socks[j].node == w->socks[i].node && w->socks[j].kind == w->socks[i].kind)
                t[j] = (uint32_t)w->socks[j].node | (((uint32_t)w->socks[j].kind | 0x80u) << 8) | (base << 16) | (K << 24);
    }
    return t;
}

inline int validate(const madsim_workload_t* w, const madsim_config_t* cfg, std::string* err) {
    if (!w || !cfg) return fail(err, MADSIM_E_ARG, "null workload/config");
    if (!w->insns || !w->progs || w->n_progs == 0 || w->n_progs > 255 || w->n_insns == 0 || w->n_insns > 4096)   /* the table is copied into LDS, 16 B per instruction */
        return fail(err, MADSIM_E_WORKLOAD, "bad program table");
    if (w->n_nodes > 31) return fail(err, MADSIM_E_WORKLOAD, "at most 31 nodes in this build");
    if (w->n_socks > 63 || (w->n_socks && !w->socks)) return fail(err, MADSIM_E_WORKLOAD, "at most 63 socket addresses");
    for (uint32_t i = 0; i < w->n_progs; i++)
        if (w->progs[i].node > w->n_nodes || w->progs[i].entry >= w->n_insns) return fail(err, MADSIM_E_WORKLOAD, "bad prog entry");
    {   // no body may run off the end of the table: the kernel fetches insns[pc] without a range check (jump targets are checked below)
        const uint8_t last = w->insns[w->n_insns - 1].op;
        if (last != MS_OP_DONE && last != MS_OP_JMP && last != MS_OP_PANIC)
            return fail(err, MADSIM_E_WORKLOAD, "the instruction table must end in MS_OP_DONE, MS_OP_JMP or MS_OP_PANIC");
    }
    for (uint32_t i = 0; i < w->n_socks; i++)
        if (w->socks[i].node == 0 || (w->socks[i].kind != MADSIM_ADDR_VIRTUAL && w->socks[i].node > w->n_nodes)) return fail(err, MADSIM_E_WORKLOAD, "bad socket node");
    for (uint32_t i = 0; i < w->n_socks; i++) {
        if (w->socks[i].kind > MADSIM_ADDR_VIRTUAL) return fail(err, MADSIM_E_WORKLOAD, "bad socket address (kind 0..3)");
        if (w->socks[i].kind == MADSIM_ADDR_VIRTUAL && w->socks[i].port == 0) return fail(err, MADSIM_E_WORKLOAD, "a virtual address needs a port");
    }
    if (w->n_services > MADSIM_MAX_SERVICES || (w->n_services && !w->services)) return fail(err, MADSIM_E_WORKLOAD, "at most 8 IPVS services");
    for (uint32_t k = 0; k < w->n_services; k++) {
        const madsim_service_t& sv = w->services[k];
        if (sv.vaddr >= w->n_socks || (sv.n_servers > 6 && sv.n_servers != MADSIM_SERVICE_ABSENT)) return fail(err, MADSIM_E_WORKLOAD, "IPVS service: bad address entry, more than 6 servers, or servers on a service declared absent");
        if (w->socks[sv.vaddr].port == 0) return fail(err, MADSIM_E_WORKLOAD, "IPVS service: the service address needs a port");
        for (uint32_t j = 0; j < (sv.n_servers & 7u); j++)
            if (sv.servers[j] >= w->n_socks || w->socks[sv.servers[j]].port == 0) return fail(err, MADSIM_E_WORKLOAD, "IPVS service: a server must be a named address entry");
        for (uint32_t j = 0; j < k; j++) {               // a HashMap keyed by the address: one service per address
            const madsim_sock_t &x = w->socks[sv.vaddr], &y = w->socks[w->services[j].vaddr];
            if (x.kind == y.kind && x.port == y.port && ((x.kind != MADSIM_ADDR_IP && x.kind != MADSIM_ADDR_VIRTUAL) || x.node == y.node))
                return fail(err, MADSIM_E_WORKLOAD, "IPVS service: two services with one address");
        }
    }
    if (w->panic_dyn_max > 254) return fail(err, MADSIM_E_WORKLOAD, "panic_dyn_max must be <= 254");
    if (device_socks(w).size() > 63) return fail(err, MADSIM_E_WORKLOAD, "at most 63 socket addresses, counting the candidate ports of ephemeral endpoints");
    for (uint32_t i = 0; i < w->n_insns; i++) {
        const madsim_insn_t& in = w->insns[i];
        switch (in.op) {
        case MS_OP_SPAWN: case MS_OP_JOIN: case MS_OP_ABORT:
            if (in.a >= w->n_progs) return fail(err, MADSIM_E_WORKLOAD, "prog operand out of range"); break;
        case MS_OP_DJNZ: case MS_OP_JMP: case MS_OP_JEQ:
            if (in.b >= w->n_insns) return fail(err, MADSIM_E_WORKLOAD, "jump target out of range"); break;
        case MS_OP_BIND: case MS_OP_REPLY: case MS_OP_RECV: case MS_OP_CLOSE: case MS_OP_RECV_TIMEOUT: case MS_OP_ACCEPT:
            if (in.a >= w->n_socks) return fail(err, MADSIM_E_WORKLOAD, "socket operand out of range");
            // (the reference would let an IP-less node bind it and answer AddrNotAvailable to every other: not modelled, refused)
            if (w->socks[in.a].kind == MADSIM_ADDR_VIRTUAL) return fail(err, MADSIM_E_WORKLOAD, "a virtual address is a destination only: it cannot be bound or used as an Endpoint");
            if ((in.op == MS_OP_REPLY || in.op == MS_OP_RECV || in.op == MS_OP_RECV_TIMEOUT) && (in.b >> 8) > MADSIM_TAG_RPC_LAST)
                return fail(err, MADSIM_E_WORKLOAD, "tags 0xFE and 0xFF are reserved");
            break;
        case MS_OP_SEND: case MS_OP_CONNECT:
            if (in.a >= w->n_socks || (uint32_t)(in.b & 0xff) >= w->n_socks) return fail(err, MADSIM_E_WORKLOAD, "socket operand out of range");
            if (w->socks[in.a].kind == MADSIM_ADDR_VIRTUAL) return fail(err, MADSIM_E_WORKLOAD, "a virtual address is never an Endpoint");
            if (in.op == MS_OP_SEND && (in.b >> 8) > MADSIM_TAG_RPC_LAST) return fail(err, MADSIM_E_WORKLOAD, "tags 0xFE and 0xFF are reserved");
            if (w->socks[in.b & 0xff].port == 0) return fail(err, MADSIM_E_WORKLOAD, "an ephemeral Endpoint (port 0) has no address a peer can name: reply to `from` or dial a named entry");
            break;
        case MS_OP_RPC_CALL:
            if (in.a >= w->n_socks || (uint32_t)(in.b & 0xff) >= w->n_socks) return fail(err, MADSIM_E_WORKLOAD, "socket operand out of range");
            if ((uint32_t)(in.b >> 8) < MADSIM_TAG_RPC_FIRST || (uint32_t)(in.b >> 8) > MADSIM_TAG_RPC_LAST) return fail(err, MADSIM_E_WORKLOAD, "rpc_call needs a typed request tag (0x80..0xFD)");
            if (w->socks[in.b & 0xff].port == 0) return fail(err, MADSIM_E_WORKLOAD, "an ephemeral Endpoint (port 0) has no address a peer can name: reply to `from` or dial a named entry");
            break;
        case MS_OP_RPC_REPLY:
            if (in.a >= w->n_socks) return fail(err, MADSIM_E_WORKLOAD, "socket operand out of range"); break;
        case MS_OP_IPVS:
            if (in.a > MADSIM_IPVS_DEL_SERVER || in.b >= w->n_services) return fail(err, MADSIM_E_WORKLOAD, "ipvs: bad call or service index");
            if (in.a >= MADSIM_IPVS_ADD_SERVER && (in.imm >= w->n_socks || w->socks[in.imm].port == 0)) return fail(err, MADSIM_E_WORKLOAD, "ipvs: a server must be a named address entry");
            break;
        case MS_OP_HOOK_REQ:
            if ((uint32_t)(in.b >> 8) < MADSIM_TAG_RPC_FIRST || (uint32_t)(in.b >> 8) > MADSIM_TAG_RPC_LAST) return fail(err, MADSIM_E_WORKLOAD, "hook_rpc_req needs a typed request tag (0x80..0xFD)");
            /* fall through */
        case MS_OP_HOOK_RSP:
        case MS_OP_BUILD: case MS_OP_KILL: case MS_OP_RESTART: case MS_OP_PAUSE: case MS_OP_RESUME:
        case MS_OP_CLOG_NODE: case MS_OP_UNCLOG_NODE: case MS_OP_ASSERT_EXIT:
            if (in.a > w->n_nodes) return fail(err, MADSIM_E_WORKLOAD, "node operand out of range"); break;
        case MS_OP_CLOG_LINK: case MS_OP_UNCLOG_LINK:
            if (in.a > w->n_nodes || in.b > w->n_nodes) return fail(err, MADSIM_E_WORKLOAD, "node operand out of range"); break;
        default: break;
        }
    }
    {   // `t0` is a local of the task body (`let t0 = Instant::now()`): Rust refuses a use before its assignment, and so does this —
        // a task slot's t0 words are whatever its previous occupant left there.  Rule: in the program that owns it (the one with the
        // largest entry <= pc) a MS_OP_MARK stands at a lower pc than every MS_OP_SLEEP_UNTIL / MS_OP_ASSERT_ELAPSED.
        std::vector<uint8_t> is_entry(w->n_insns, 0);
        for (uint32_t p = 0; p < w->n_progs; p++) is_entry[w->progs[p].entry] = 1;
        bool marked = false;
        for (uint32_t i = 0; i < w->n_insns; i++) {
            if (is_entry[i]) marked = false;
            const uint8_t op = w->insns[i].op;
            if (op == MS_OP_MARK) marked = true;
            else if ((op == MS_OP_SLEEP_UNTIL || op == MS_OP_ASSERT_ELAPSED) && !marked)
                return fail(err, MADSIM_E_WORKLOAD, "sleep_until / assert_elapsed before the program's first mark: t0 is not assigned yet");
        }
    }
    for (uint32_t p = 0; p < w->n_progs; p++) {
        if (!(w->progs[p].flags & MADSIM_PROG_DROP_SPAWN)) continue;
        if (p + 1 >= w->n_progs || w->progs[p + 1].node != w->progs[p].node)
            return fail(err, MADSIM_E_WORKLOAD, "MADSIM_PROG_DROP_SPAWN: the guard spawns program p + 1, which must exist and run on the same node");
        if (uses_op(w, MS_OP_PAUSE))
            return fail(err, MADSIM_E_WORKLOAD, "MADSIM_PROG_DROP_SPAWN cannot be combined with MS_OP_PAUSE (a parked Runnable is dropped in the killer's context)");
    }
    {   // NetSim::reset_node (network.rs:142-147) drops a killed node's sockets in the iteration order of a HashMap hashed with
        // the seed's SipHash keys (rand.rs:176-180) — observable only when two of them hold un-accepted connections at that moment
        // (each drop closes queued connection ends and wakes their peers: the wake order feeds the ready queue).  That order is
        // not modelled, so a workload where it COULD matter is refused rather than answered differently: a node that can be
        // reset (kill / restart / restart_on_panic / an init task, whose return exits the node) with two or more listening entries.
        uint32_t resettable = 0;
        for (uint32_t i = 0; i < w->n_insns; i++) {
            const madsim_insn_t& in = w->insns[i];
            if (in.op == MS_OP_KILL || in.op == MS_OP_RESTART) resettable |= 1u << in.a;
        }
        for (uint32_t n = 0; n <= w->n_nodes && w->nodes; n++) if (w->nodes[n].flags & (MADSIM_NODE_RESTART_ON_PANIC | MADSIM_NODE_RESTART_MATCHING)) resettable |= 1u << n;
        for (uint32_t p = 0; p < w->n_progs; p++) if (w->progs[p].flags & MADSIM_PROG_INIT) resettable |= 1u << w->progs[p].node;
        for (uint32_t n = 1; n <= w->n_nodes; n++) {
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < w->n_insns; i++) {
                const madsim_insn_t& in = w->insns[i];
                if (in.op != MS_OP_ACCEPT || w->socks[in.a].node != n) continue;
                bool seen = false;
                for (uint32_t j = 0; j < i; j++) seen |= w->insns[j].op == MS_OP_ACCEPT && w->insns[j].a == in.a;
                cnt += !seen;
            }
            if (((resettable >> n) & 1) && cnt >= 2)
                return fail(err, MADSIM_E_WORKLOAD, "a node that can be killed or restarted has two listening (accept1) Endpoints: the order in which "
                                                    "reset_node drops their queued connections (a seeded HashMap, network.rs:142-147) is not modelled");
        }
    }
    if (cfg->lat_lo_ns >= cfg->lat_hi_ns) return fail(err, MADSIM_E_ARG, "send_latency: cannot sample empty range");
    if (!(cfg->packet_loss_rate >= 0.0 && cfg->packet_loss_rate <= 1.0)) return fail(err, MADSIM_E_ARG, "packet_loss_rate not in [0,1]");
    // MS_OP_SET_LATENCY: `gen_range` of an empty send_latency panics at the next link test (network.rs:267): a table entry must be a range
    if (cfg->n_lat_table > 4) return fail(err, MADSIM_E_ARG, "lat_table holds at most 4 entries");
    for (uint32_t k = 0; k < cfg->n_lat_table; k++)
        if (cfg->lat_table_lo_ns[k] >= cfg->lat_table_hi_ns[k]) return fail(err, MADSIM_E_ARG, "lat_table: cannot sample empty range");
    for (uint32_t i = 0; i < w->n_insns; i++)
        if (w->insns[i].op == MS_OP_SET_LATENCY && w->insns[i].a >= cfg->n_lat_table)
            return fail(err, MADSIM_E_WORKLOAD, "set_latency names an entry beyond madsim_config_t.n_lat_table");
    return 0;
}

// The config a geometry query runs with (madsim_hip_geometry has no config argument): Config::default() and a full latency table, so
// that a workload with MS_OP_SET_LATENCY validates — the layout does not depend on what the entries hold.
inline madsim_config_t probe_config() {
    madsim_config_t cfg{}; cfg.lat_lo_ns = 1000000; cfg.lat_hi_ns = 10000000;
    cfg.n_lat_table = 4;
    for (int i = 0; i < 4; i++) { cfg.lat_table_lo_ns[i] = 1000000; cfg.lat_table_hi_ns[i] = 10000000; }
    return cfg;
}

// buckets of the re-registration table (a power of two; k_timer.h dedup_note): a seed's share of the launch's working set
#ifndef MADSIM_DEDUP_BUCKETS
#define MADSIM_DEDUP_BUCKETS 64
#endif
static_assert(MADSIM_DEDUP_BUCKETS <= 64 && (MADSIM_DEDUP_BUCKETS & (MADSIM_DEDUP_BUCKETS - 1)) == 0, "Lane::dd_occ mirrors the buckets in 64 bits");
inline int make_geometry(const Device& g, const madsim_workload_t* w, const madsim_config_t* cfg, const madsim_limits_t* lim, uint64_t count, Geo* G, std::string* err, bool trace = false) {
    KParams& P = G->P;
    memset(&P, 0, sizeof P);
    madsim_limits_t L{};
    if (lim) L = *lim;
    P.n_insns = w->n_insns; P.n_progs = w->n_progs; P.n_nodes = w->n_nodes;
    P.n_socks = (uint32_t)device_socks(w).size();        // the caller's entries + the candidate ports of ephemeral endpoints
    P.uses_eph = P.n_socks != w->n_socks;
    bernoulli(cfg->packet_loss_rate, &P.loss_pint, &P.loss_always);
    P.buggify = cfg->buggify != 0;
    uint32_t dummy; bernoulli(0.1, &P.bug_pint, &dummy);
    uniform_duration_params(cfg->lat_lo_ns, cfg->lat_hi_ns, &P.lat_mode, &P.lat_low, &P.lat_range, &P.lat_zone);
    for (int i = 0; i < 4; i++) bernoulli(i < (int)cfg->n_loss_table ? cfg->loss_table[i] : 0.0, &P.loss_table_pint[i], &P.loss_table_always[i]);
    P.uses_set_lat = uses_op(w, MS_OP_SET_LATENCY);
    for (uint32_t i = 0; i < 4 && i < cfg->n_lat_table; i++)
        uniform_duration_params(cfg->lat_table_lo_ns[i], cfg->lat_table_hi_ns[i], &P.lat_tab_mode[i], &P.lat_tab_low[i], &P.lat_tab_range[i], &P.lat_tab_zone[i]);
    P.time_limit = L.time_limit_ns;
    P.max_steps = L.max_steps ? L.max_steps : (1u << 24);
    P.no_log = L.no_trace_hash ? 1u : 0u;
    bool restarts = uses_op(w, MS_OP_RESTART);
    for (uint32_t i = 0; i <= w->n_nodes && w->nodes; i++) restarts |= (w->nodes[i].flags & (MADSIM_NODE_RESTART_ON_PANIC | MADSIM_NODE_RESTART_MATCHING)) != 0;
    // request-per-connection servers spawn a handler per accept: leave room for a few concurrent ones
    bool chan = uses_op(w, MS_OP_CONNECT) || uses_op(w, MS_OP_ACCEPT);
    P.uses_hooks = uses_op(w, MS_OP_HOOK_REQ) || uses_op(w, MS_OP_HOOK_RSP);
    P.uses_rpc = uses_op(w, MS_OP_RPC_CALL) || uses_op(w, MS_OP_RPC_REPLY) || P.uses_hooks;   // likewise: one task per request in flight
    P.max_tasks = L.max_tasks ? L.max_tasks : w->n_progs + (restarts ? w->n_progs : 0) + (chan || P.uses_rpc ? 8 : 0);
    if (P.max_tasks > 254) P.max_tasks = 254;
    if (P.max_tasks > 254) return fail(err, MADSIM_E_LIMITS, "max_tasks must be <= 254");
    P.mbox_regs = L.mbox_regs == MADSIM_LIMIT_NONE ? 0 : L.mbox_regs ? L.mbox_regs : 2;
    P.mbox_msgs = L.mbox_msgs == MADSIM_LIMIT_NONE ? 0 : L.mbox_msgs ? L.mbox_msgs : 2;
    if (P.mbox_regs > 255 || P.mbox_msgs > 255) return fail(err, MADSIM_E_LIMITS, "mailbox capacities must be <= 255");
    P.heap_lds = L.heap_lds_slots ? L.heap_lds_slots : 8;
    P.heap_spill = (L.heap_lds_slots || L.heap_spill_slots) ? L.heap_spill_slots : 56;
    bool t0 = uses_op(w, MS_OP_MARK) || uses_op(w, MS_OP_SLEEP_UNTIL) || uses_op(w, MS_OP_ASSERT_ELAPSED) || uses_op(w, MS_OP_RECV_TIMEOUT) || P.uses_rpc;
    // (MS_OP_CCLOSE alone counts too: without the connection unit a stray `drop((tx, rx))` read the task's flag word as a connection id)
    P.uses_chan = uses_op(w, MS_OP_CONNECT) || uses_op(w, MS_OP_ACCEPT) || uses_op(w, MS_OP_CSEND) || uses_op(w, MS_OP_CRECV) || uses_op(w, MS_OP_CCLOSE);
    // task units: 0-1 always; 2 = {t0, timeout()'s deadline} when used; then the connection unit, then the RPC unit
    P.task_units = t0 ? 3 : 2;
    if (P.uses_chan) { P.chan_unit = P.task_units; P.task_units++; }
    // {rsp_tag in hand, rsp_tag staged with the oneshot value}: the two t0 words of unit 2 when no MARK-family op needs
    // them (unit 2's other half is the timeout deadline RPC calls use anyway), else a unit of their own
    const bool mark = uses_op(w, MS_OP_MARK) || uses_op(w, MS_OP_SLEEP_UNTIL) || uses_op(w, MS_OP_ASSERT_ELAPSED);
    if (P.uses_rpc) { if (!mark) P.rpc_unit = 2; else { P.rpc_unit = P.task_units; P.task_units++; } }
    // per socket: header, owner, registrations, queued messages (+ accept queue, parked acceptor); set once the layout
    // (base or extended) is known, below
    P.max_conns = L.max_conns ? L.max_conns : 4;
    P.chan_queue = L.chan_queue ? L.chan_queue : 2;
    if (P.max_conns > 127 || P.chan_queue > 15) return fail(err, MADSIM_E_LIMITS, "max_conns <= 127, chan_queue <= 15");
    P.conn_words = 3 + 2 * P.chan_queue * 3;
    P.has_clog_link = uses_op(w, MS_OP_CLOG_LINK) || uses_op(w, MS_OP_UNCLOG_LINK);
    P.has_clog = P.has_clog_link || uses_op(w, MS_OP_CLOG_NODE) || uses_op(w, MS_OP_UNCLOG_NODE);
    // uniq_addr: every table entry is a distinct node-IP address and every node has its IP — then an address resolves to
    // its node and to its own table entry, and the kernel skips the general resolution of network.rs:272-313
    P.uniq_addr = 1;
    for (uint32_t i = 0; i < w->n_socks; i++) {
        if (w->socks[i].kind != MADSIM_ADDR_IP || w->socks[i].port == 0) P.uniq_addr = 0;      // (virtual addresses included)
        for (uint32_t j = i + 1; j < w->n_socks; j++)
            if (w->socks[i].node == w->socks[j].node && w->socks[i].port == w->socks[j].port) P.uniq_addr = 0;
    }
    for (uint32_t i = 0; i <= w->n_nodes && w->nodes; i++) if (w->nodes[i].flags & MADSIM_NODE_NO_IP) P.uniq_addr = 0;
    if (w->n_services) P.uniq_addr = 0;                // IPVS rewrites destinations: general address resolution
    // the node table: a flags word per node, then the restart rows (when a node restarts on matching panics), then the services
    {
        bool rows = false;
        for (uint32_t i = 0; i <= w->n_nodes && w->nodes; i++) rows |= (w->nodes[i].flags & MADSIM_NODE_RESTART_MATCHING) != 0;
        P.n_nodetab = w->n_nodes + 1;
        P.pm_off = rows ? P.n_nodetab : 0;
        if (rows) P.n_nodetab += 8 * (w->n_nodes + 1);
        P.n_services = w->n_services;
        P.svc_off = w->n_services ? P.n_nodetab : 0;
        P.n_nodetab += 2 * w->n_services;
        P.panic_dyn_max = w->panic_dyn_max ? w->panic_dyn_max : 254u;
    }
    // ---- which optional per-seed regions exist (LDS diet: a workload only carries what it can touch) ----
    P.restart_nodes = 0;
    for (uint32_t i = 0; i <= w->n_nodes && w->nodes; i++)
        if (w->nodes[i].flags & (MADSIM_NODE_RESTART_ON_PANIC | MADSIM_NODE_RESTART_MATCHING)) P.restart_nodes |= 1u << i;
    P.has_restart_on_panic = P.restart_nodes != 0;
    // Classes of extended ops the workload needs (sim_kernel.h MADSIM_FEAT_*): the kernel build is picked by this mask
    // (select_variant), and any of them switches the per-seed LDS layout to its extended form.
    P.features = 0;
    if (uses_op(w, MS_OP_RECV_TIMEOUT) || uses_op(w, MS_OP_MARK) || uses_op(w, MS_OP_SLEEP_UNTIL) || uses_op(w, MS_OP_ASSERT_ELAPSED) ||
        uses_op(w, MS_OP_ADVANCE) || uses_op(w, MS_OP_TRACE_TIME) || P.uses_set_lat) P.features |= MADSIM_FEAT_TIME;   // (set_latency: any extended build; this is the leanest)
    if (P.uses_chan) P.features |= MADSIM_FEAT_CHAN;
    if (P.uses_rpc) P.features |= MADSIM_FEAT_RPC | MADSIM_FEAT_TIME;            // call_timeout rides the timeout unit
    if (P.has_restart_on_panic || uses_op(w, MS_OP_KILL) || uses_op(w, MS_OP_RESTART) || uses_op(w, MS_OP_PAUSE) || uses_op(w, MS_OP_RESUME) ||
        uses_op(w, MS_OP_ABORT) || uses_op(w, MS_OP_ASSERT_EXIT) || uses_op(w, MS_OP_BUILD)) P.features |= MADSIM_FEAT_NODE;
    for (uint32_t i = 0; i < w->n_progs; i++) if (w->progs[i].flags & MADSIM_PROG_INIT) P.features |= MADSIM_FEAT_NODE;
    if (!P.uniq_addr) P.features |= MADSIM_FEAT_ADDR;
    for (uint32_t i = 0; i < w->n_progs; i++)          // guards that spawn in Drop: compiled into the full builds only
        if (w->progs[i].flags & MADSIM_PROG_DROP_SPAWN) P.features |= MADSIM_FEAT_NODE | MADSIM_FEAT_ADDR;
    if (trace) P.features = MADSIM_FEAT_ALL;          // the trace build carries every class
    P.lifecycle = P.features != 0;
    const uint32_t cus = g.num_cus > 0 ? (uint32_t)g.num_cus : 256u;
    uint32_t lw = 64;
    uint32_t sh_bytes = 0;
    // Where the task table and the planes live (madsim_limits_t.state_mem): LDS, or — extended-op workloads whose state
    // would leave a CU with fewer than four full waves — global memory, [unit][lane] across the launch (Variant::G, k_state.h).
    // (its low byte; MADSIM_STATE_DEDUP_TIMERS rides above it)
    const uint32_t state_mem = L.state_mem & 0xffu;
    if (state_mem > MADSIM_STATE_COMPACT || (L.state_mem & ~(0xffu | MADSIM_STATE_DEDUP_TIMERS | MADSIM_STATE_NARROW_HEAP)))
        return fail(err, MADSIM_E_LIMITS, "state_mem must be 0 (auto), 1 (LDS), 2 (global) or 3 (compact), optionally | MADSIM_STATE_DEDUP_TIMERS | MADSIM_STATE_NARROW_HEAP");
    // MADSIM_STATE_NARROW_HEAP: 8-byte heap entries hold the low deadline word — admitted when nothing the workload can ask for lies 2^31 ns
    // ahead of the clock (the device checks every push all the same: a channel back-off can grow past it at run time)
    bool narrow_ok = (L.state_mem & MADSIM_STATE_NARROW_HEAP) && !trace && !cfg->buggify && !P.has_restart_on_panic;
    {
        uint64_t horizon = std::max<uint64_t>(cfg->lat_hi_ns, 1000000ull);
        for (uint32_t i = 0; i < cfg->n_lat_table && i < 4; i++) horizon = std::max<uint64_t>(horizon, cfg->lat_table_hi_ns[i]);
        for (uint32_t i = 0; i < w->n_insns; i++) {
            const madsim_insn_t& in = w->insns[i];
            if (in.op == MS_OP_SLEEP || in.op == MS_OP_SLEEP_RAND || in.op == MS_OP_SLEEP_UNTIL || in.op == MS_OP_ADVANCE)
                horizon = std::max<uint64_t>(horizon, (uint64_t)in.b * 1000000000ull + in.imm);
            if (in.op == MS_OP_RECV_TIMEOUT) horizon = std::max<uint64_t>(horizon, (uint64_t)(in.b & 0xff) * 1000000000ull + in.imm);
            if (in.op == MS_OP_RPC_CALL) horizon = std::max<uint64_t>(horizon, (uint64_t)(in.imm >> 8) * 1000000ull);
        }
        if (horizon >= (1ull << 31) - (1ull << 24)) narrow_ok = false;
    }
    P.narrow = 0; P.pool_n = 0; P.pool_off = 0; P.off_pmask = 0;
    bool use_narrow = false;
    // delivery records of a narrow heap: a quarter of the heap's capacity (in-flight datagrams are a fifth of its entries), 32 .. 256
    auto pool_records = [](uint32_t heap_cap) { return std::min<uint32_t>(256u, std::max<uint32_t>(32u, (heap_cap / 4 + 31) & ~31u)); };
    P.gstate_mode = 0;
    P.dedup_n = 0; P.dedup_off = 0;
    // base-op builds: no owner word per socket (the owner's slot rides in the header, k_state.h) and 8-byte unit1
    P.sock_words = (P.lifecycle ? 2 : 1) + P.mbox_regs + 2 * P.mbox_msgs + (P.uses_chan ? 3 : 0);   // + accept queue (2 words), parked acceptor
    if (!P.lifecycle && P.mbox_msgs > 127) return fail(err, MADSIM_E_LIMITS, "mbox_msgs must be <= 127 for workloads without extended ops");
    const uint32_t task_bytes = P.lifecycle ? P.task_units * 16 : 24;
    const uint32_t heap_bytes = P.lifecycle ? 16 : 12;        // base ops: {deadline 8, meta 4}, no payload word (k_timer.h)
    for (int pass = 0; pass < 3; pass++) {
        if (!P.lifecycle && P.max_tasks < P.n_progs) P.max_tasks = P.n_progs;   // handle words live in task slots there
        // the ready queue lives in a register in the (base ops, <= 8 tasks, full 64-lane waves) builds: lay out with it
        // first and once more without it if the lane stride does not come out at 64 (select_variant)
        P.rq_in_reg = !P.lifecycle && P.max_tasks <= 8 && !trace && pass == 0;
        P.off_ready = 0;
        P.off_socks = P.gstate_mode ? 0 : P.off_ready + (P.rq_in_reg ? 0 : P.max_tasks);     // global planes start at the sockets
        P.off_handles = P.off_socks + P.n_socks * P.sock_words;
        // JoinHandle words: a plane with the extended ops, else unit1.y of task slot p (sim_kernel.hip HW)
        P.off_nodes = P.off_handles + (P.lifecycle ? P.n_progs : 0);
        // node region (extended ops only): killed / paused / gen0_killed masks, spawn counter, one info_gen byte per node
        P.off_clog = P.off_nodes + (P.lifecycle ? 4 + (P.n_nodes + 4) / 4 + 1 : 0);   // + the base-time word
        P.off_pause = P.off_clog + (P.has_clog ? 2 + (P.has_clog_link ? P.n_nodes + 1 : 0) : 0);
        P.uses_pause = uses_op(w, MS_OP_PAUSE);
        P.off_greg = P.off_pause + (P.uses_pause ? 1 + P.max_tasks : 0);
        bool gregs = uses_op(w, MS_OP_GSET) || uses_op(w, MS_OP_GADD) || uses_op(w, MS_OP_ASSERT_G) || uses_op(w, MS_OP_PANIC_IF_G_LT);
        for (uint32_t i = 0; i < w->n_insns; i++) gregs |= w->insns[i].op == MS_OP_PANIC && (w->insns[i].a & 1);   // panic!("{}", flag)
        P.off_conn = P.off_greg + (gregs ? 4 : 0);
        P.off_hooks = P.off_conn + (P.uses_chan ? P.max_conns * P.conn_words : 0);
        P.off_ipvs = P.off_hooks + (P.uses_hooks ? P.n_nodes + 1 : 0);
        P.ipvs_dyn = uses_op(w, MS_OP_IPVS);
        P.lane_words = P.off_ipvs + P.n_services * (P.ipvs_dyn ? 2 : 1);
        if (P.gstate_mode) {           // the planes just laid out go to the global block; LDS keeps the ready queue only
            P.gs_plane_words = P.lane_words;
            // MADSIM_STATE_DEDUP_TIMERS: 64 buckets of 16 bytes behind the task units — the one build that carries the code is
            // the global-state build of timeout-only workloads (k_state.h Variant::DEDUP, k_timer.h dedup_note)
            // the task region: one granule per (slot, lane), the slot's units side by side (k_state.h gs_addr_task)
            P.gs_gran_sh = 5; while ((1u << P.gs_gran_sh) < P.task_units * 16) P.gs_gran_sh++;
            P.dedup_off = P.max_tasks << P.gs_gran_sh;
            // (tried in round 4: the every-class build with the switch — the topology's heap holds 68 entries of which 30 are distinct —
            // bit-exact, but the build, already 24 registers short, lost 3 % with the code compiled in and 10 % with it switched on)
            P.dedup_n = ((L.state_mem & MADSIM_STATE_DEDUP_TIMERS) && P.features == MADSIM_FEAT_TIME && !trace) ? (uint32_t)MADSIM_DEDUP_BUCKETS : 0u;
            P.gs_planes = P.dedup_off + P.dedup_n * 16;
            P.gs_stride = (P.gs_planes + P.gs_plane_words * 4 + 63) & ~63u;
            P.off_amask = (P.max_tasks + 3) / 4;               // LDS planes: ready queue (a byte per entry), alive-task mask, owned-socket mask
            P.off_omask = P.off_amask + (P.max_tasks + 31) / 32;
            P.lane_words = P.off_omask + 2;
            // narrow heap entries, where the layout has a spill region and a build with the variant exists (sim_kernel.h): the delivery
            // records — a quarter of the heap's capacity, in-flight datagrams are a fifth of its entries — behind the planes
            P.narrow = use_narrow ? 1u : 0u;                    // (decided where the LDS quota was split, below)
            P.pool_n = use_narrow ? pool_records(P.heap_lds + P.heap_spill) : 0u;
            if (P.narrow) {
                P.pool_off = P.gs_stride;
                P.gs_stride += P.pool_n * 8;
                P.off_pmask = P.off_omask + 2;
                P.lane_words = P.off_pmask + P.pool_n / 32;
            }
        }
        P.sh_insns = 0;
        P.sh_progs = P.sh_insns + 4 * P.n_insns;
        P.sh_socks = P.sh_progs + P.n_progs;
        P.sh_nodes = P.sh_socks + P.n_socks;
        P.sh_heap = (P.sh_nodes + P.n_nodetab + 3) & ~3u;
        sh_bytes = P.sh_heap * 4;
        G->lds_per_seed = P.heap_lds * (P.narrow ? 8u : heap_bytes) + (P.gstate_mode ? 0 : P.max_tasks * task_bytes) + P.lane_words * 4;
        if (sh_bytes + 8 * (size_t)G->lds_per_seed > g.lds_per_cu && (state_mem == MADSIM_STATE_LDS || !P.lifecycle || trace || P.gstate_mode)) return fail(err, MADSIM_E_LIMITS, "per-seed LDS state too large: lower heap_lds_slots / mailbox capacities");
        // Lanes per wave (lw): how many of a wave's 64 lanes carry a seed.  Measured on MI355X (4-node
        // ping-pong, 65 536 seeds, profiles/r1_lanes_per_wave.md): every wave-instruction costs the SIMD
        // ~4 cycles whatever the number of active lanes, and at 16 lanes/wave the VALU pipe is already
        // ~98 % busy, so trading lanes for more waves loses (64: 5.36 ms, 32: 6.08, 16: 8.32, 8: 16.7).
        // Full waves are the default; the knob stays for experiments (madsim_limits_t.lanes_per_wave).
        lw = 64;
        if (L.lanes_per_wave) {
            lw = L.lanes_per_wave;
            if (lw != 8 && lw != 16 && lw != 32 && lw != 64) return fail(err, MADSIM_E_LIMITS, "lanes_per_wave must be 8, 16, 32 or 64");
        } else {
            // Large per-seed state: a CU has 4 SIMDs and a wave runs on one of them, so when LDS admits fewer than four
            // full waves per CU, carry fewer seeds per wave until at least four workgroups fit (one per SIMD) — the same
            // seeds in flight, spread over all SIMDs.  (When >= 4 full waves fit, full waves win: r1_lanes_per_wave.md.)
            auto blocks = [&](uint32_t l) { size_t b = (size_t)sh_bytes + (size_t)l * G->lds_per_seed; return b > g.lds_per_cu ? 0u : (uint32_t)(g.lds_per_cu / b); };
            while (lw > 8 && blocks(lw) < 4) lw >>= 1;
        }
        if (P.gstate_mode) { if (!L.lanes_per_wave) lw = 64; break; }
        // extended-op workloads only: a base-op ping-pong iteration is ~11k cycles and 26 G iterations/s would need
        // > 15 TB/s of 64-byte sector traffic (measured: 3.2 ms per batch against 1.99 ms LDS-resident, profiles/r2_experiments.md)
        const bool can_g = P.lifecycle && !trace && state_mem != MADSIM_STATE_LDS;
        if (can_g && (state_mem == MADSIM_STATE_GLOBAL || (lw != 64 && !L.lanes_per_wave))) {
            P.gstate_mode = 1;
            // LDS now holds little more than the top of the timer heap.  Keep as much of the requested LDS quota as still
            // lets the build's register budget decide the occupancy — as many 4-wave workgroups per CU as the build fits waves
            // per SIMD (three at <= 168 VGPRs, two above) — each workgroup with its copy of the tables.  Every level
            // that stays in LDS is one global round trip less per sift, but with the [unit][lane] state layout a third wave
            // per SIMD is worth more (election loop: 8 entries x 12 waves 6.5 G steps/s, 15 x 8 6.1).  The rest of the
            // quota moves to the coalesced spill region, same capacity.
            const madsim_k::VariantSel gsel = madsim_k::select_variant(P, trace);
            const int gv = g.vgprs ? g.vgprs(&gsel) : -1;
            uint32_t per_simd = gv > 0 ? 512u / (uint32_t)((gv + 7) & ~7) : (gsel.feat & (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR)) != (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR) ? 3u : 2u;
            per_simd = per_simd < 2 ? 2u : per_simd > 4 ? 4u : per_simd;         // workgroups of four waves, one wave per SIMD each
            if (g.max_waves_per_simd > 0 && per_simd > (uint32_t)g.max_waves_per_simd) per_simd = (uint32_t)g.max_waves_per_simd;
            const size_t quota = g.lds_per_cu / per_simd;                    // (a large instruction table can eat a workgroup's whole share:
            const size_t glw = L.lanes_per_wave ? lw : 64;                   // (auto: full waves, whatever the LDS-resident sizing above tried)
            const size_t per_seed = quota > (size_t)sh_bytes + 1280 ? (quota - sh_bytes - 1280) / (4 * glw) : 0;   // no size_t underflow)
            const size_t fixed = 4 * (((size_t)P.max_tasks + 3) / 4 + (P.max_tasks + 31) / 32 + 2);
            uint32_t fit = per_seed > fixed + 64 ? (uint32_t)((per_seed - fixed) / 16) : 4u;
            // Narrow heap entries (MADSIM_STATE_NARROW_HEAP) where the 16-byte layout would spill and a build with the variant exists:
            // 8 bytes per entry and the record pool's mask words in the same quota — what then fits may well be the whole heap
            use_narrow = false;
            if (narrow_ok && (P.heap_spill || P.heap_lds > fit)) {
                KParams Q = P; Q.narrow = 1; Q.gstate_mode = 1; Q.lw_shift = glw == 32 ? 5 : 6;
                if (!Q.heap_spill) Q.heap_spill = 1;
                use_narrow = madsim_k::variant_compiled(madsim_k::select_variant(Q, trace));
            }
            if (use_narrow) {
                const size_t fixed_n = fixed + 4 * (pool_records(P.heap_lds + P.heap_spill) / 32);
                fit = per_seed > fixed_n + 64 ? (uint32_t)((per_seed - fixed_n) / 8) : 4u;
            }
            if (P.heap_lds > fit) { P.heap_spill += P.heap_lds - fit; P.heap_lds = fit; }
            continue;
        }
        if (lw == 64 || !P.rq_in_reg) break;
    }
    // (32 seed lanes per wave: half the seeds share a CU's LDS — twice the timer-heap entries per seed stay out of the spill region —
    // and the launch's working set halves; the election loop keeps 0.9 of its rate with every other lane idle, round 4)
    if (P.gstate_mode && lw != 64 && !(lw == 32 && P.features == MADSIM_FEAT_TIME))
        return fail(err, MADSIM_E_LIMITS, "global state (state_mem = 2) runs full 64-lane waves; 32 seed lanes per wave for timeout-only workloads");
    P.lw_shift = lw == 8 ? 3 : lw == 16 ? 4 : lw == 32 ? 5 : 6;
    // The compact base-op layout (sim_kernel.h MADSIM_FEAT_COMPACT): 8-byte heap entries with the root in registers, the main
    // task's 24 bytes in global memory.  Taken when it buys a CU another 4-wave workgroup (the 4-node ping-pong: 200 -> 152 bytes
    // per seed, three -> four waves per SIMD) and every live deadline provably stays within 2^31 ns of the clock: the longest
    // sleep of the program, the latency range, rand_delay's 1 ms floor — or buggify's 1..5 s, which rules it out.
    P.compact = 0;
    uint32_t heap_n = P.heap_lds, heap_b = P.narrow ? 8u : heap_bytes, task_n = P.max_tasks;
    {
        uint64_t horizon = std::max<uint64_t>(cfg->lat_hi_ns, 1000000ull);
        if (cfg->buggify) horizon = ~0ull;
        for (uint32_t i = 0; i < w->n_insns; i++) {
            const madsim_insn_t& in = w->insns[i];
            if (in.op == MS_OP_SLEEP || in.op == MS_OP_SLEEP_RAND)
                horizon = std::max<uint64_t>(horizon, (uint64_t)in.b * 1000000000ull + in.imm);
        }
        const bool can = !P.lifecycle && !trace && !P.gstate_mode && lw == 64 && P.rq_in_reg && P.heap_spill == 0 && P.heap_lds >= 2 &&
                         P.max_tasks >= 2 && horizon < (1ull << 31) - (1ull << 24) && !L.lanes_per_wave;
        if (state_mem == MADSIM_STATE_COMPACT && !can)
            return fail(err, MADSIM_E_LIMITS, "state_mem = 3 (compact): base-op workloads on full waves with <= 8 tasks, no heap spill, no buggify and sleeps below 2.1 s only");
        if (can && state_mem != MADSIM_STATE_LDS) {
            const size_t per_seed_c = (size_t)(P.heap_lds - 1) * 8 + (size_t)(P.max_tasks - 1) * task_bytes + (size_t)P.lane_words * 4;
            auto groups = [&](size_t per_seed) { size_t b = ((size_t)sh_bytes + 256 * per_seed + 1279) / 1280 * 1280; return g.lds_per_cu / b; };
            if (state_mem == MADSIM_STATE_COMPACT || (groups(per_seed_c) > groups(G->lds_per_seed) && groups(per_seed_c) <= 4)) {
                P.compact = 1; heap_n = P.heap_lds - 1; heap_b = 8; task_n = P.max_tasks - 1;
                G->lds_per_seed = (uint32_t)per_seed_c;
                P.gs_stride = 24;                      // the main task's record: unit0 (16 bytes) + unit1 {x, y}
            }
        }
    }
    P.sh_tasks = P.sh_heap + heap_n * lw * (heap_b / 4);
    P.sh_planes = P.sh_tasks + (P.gstate_mode ? 0 : task_n * (task_bytes / 4) * lw);
    P.wave_words = P.sh_planes + P.lane_words * lw - P.sh_heap;
    if ((size_t)(P.sh_heap + P.wave_words) * 4 > g.lds_per_cu) return fail(err, MADSIM_E_LIMITS, "per-wave LDS exceeds 160 KiB: lower heap_lds_slots / mailbox capacities");
    // Workgroup = W independent waves.  Measured on MI355X (tools/placement.hip, profiles/r1_placement.txt): the
    // dispatcher spreads the waves of a 256-thread workgroup one per SIMD and keeps three or more such launches
    // co-resident and balanced, while one-wave workgroups stop overlapping beyond two concurrent launches.  So take
    // the largest W in {4, 2, 1} that costs at most one wave of LDS occupancy per CU.
    const uint64_t want = (count + lw - 1) / lw;      // waves this batch needs
    // LDS is handed out in 1 280-byte granules (measured, tools/placement.hip: three 53 688-byte workgroups share a CU,
    // three 54 000-byte ones do not), so a workgroup costs its size rounded up to that.
    auto lds_alloc = [&](uint32_t w2) { size_t b = (size_t)(P.sh_heap + w2 * P.wave_words) * 4; return (b + 1279) / 1280 * 1280; };
    // VGPR budget (tools/kernel_meta.sh): base builds ~110 VGPRs = 4 waves per SIMD, single-class builds 133 / 151 = 3,
    // the full extended build ~180 = 2
    const madsim_k::VariantSel vsel = madsim_k::select_variant(P, trace);
    uint32_t cap = (vsel.feat & MADSIM_FEAT_ALL) == 0 ? 16u : (vsel.feat == MADSIM_FEAT_TIME || vsel.feat == MADSIM_FEAT_CHAN || (vsel.g && (vsel.feat & (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR)) != (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR))) ? 12u : 8u;
    if (g.vgprs) {                       // 512 VGPRs per SIMD lane, allocated in blocks of 8; at most 8 waves per SIMD
        int r = g.vgprs(&vsel);
        if (r > 0) { uint32_t per_simd = 512u / (uint32_t)((r + 7) & ~7); cap = 4u * (per_simd > 8 ? 8u : per_simd < 1 ? 1u : per_simd); }
    }
    auto waves_at = [&](uint32_t w2) {
        uint32_t blocks = (uint32_t)(g.lds_per_cu / lds_alloc(w2));
        return blocks * w2 < cap ? blocks * w2 : cap;
    };
    const uint32_t best = std::max(waves_at(1), std::max(waves_at(2), waves_at(4)));
    uint32_t W = 1;
    for (uint32_t w2 = 4; w2 >= 1; w2 >>= 1)        // powers of two: a 1 024-wave batch then fills whole workgroups
        if (waves_at(w2) + (best >= 8 ? 1u : 0u) >= best && waves_at(w2) > 0) { W = w2; break; }   // largest W within one wave of the best (none to spare below 8)
    if (want < W) W = want > 1 ? 2 : 1;
    P.waves_per_block = W;
    G->lds_bytes = (P.sh_heap + W * P.wave_words) * 4;
    uint32_t bpc = (uint32_t)(g.lds_per_cu / lds_alloc(W));
    if (bpc * W > cap) bpc = cap / W;
    if (bpc == 0) bpc = 1;
    G->blocks_per_cu = bpc;
    G->waves_per_block = W;
    G->lanes_per_wave = lw;
    uint64_t want_blocks = (want + W - 1) / W;
    uint64_t resident = (uint64_t)bpc * cus;
    G->grid = (uint32_t)(want_blocks < resident ? want_blocks : resident);
    if (G->grid == 0) G->grid = 1;
    P.total_lanes = G->grid * W * lw;
    // the pair (parameter block, build) and the LDS arithmetic are checked once more as a whole: an inconsistent pair is refused here,
    // it is never launched (sim_kernel.h variant_mismatch; tests/test_geometry_consistency.py walks the combinations)
    if (const char* why = madsim_k::variant_mismatch(P, madsim_k::select_variant(P, trace), trace))
        return fail(err, MADSIM_E_LIMITS, std::string("no kernel build for this geometry: ") + why);
    if ((size_t)G->lds_bytes > g.lds_per_cu || (size_t)G->blocks_per_cu * lds_alloc(W) > std::max<size_t>(g.lds_per_cu, lds_alloc(W)))
        return fail(err, MADSIM_E_LIMITS, "no kernel build for this geometry: the workgroups of a CU exceed its LDS");
    return 0;
}


// Device form of the workload tables.  MS_OP_SLEEP_RAND's `a` (lo in 50 ms units) is replaced by an index
// into `durs` = {mode, low, range, zone} of UniformDuration::new(lo, hi) [DEP rand 0.8, SURVEY A.3].
struct DeviceTables { std::vector<uint32_t> insns, progs, socks, nodes; std::vector<uint64_t> durs; };
inline int build_tables(const madsim_workload_t* w, DeviceTables* T, std::string* err) {
    T->insns.resize(4 * (size_t)w->n_insns); T->progs.resize(w->n_progs);
    T->socks = device_socks(w);
    if (T->socks.empty()) T->socks.push_back(0);
    T->durs.assign(4, 0);
    for (uint32_t i = 0; i < w->n_insns; i++) {
        madsim_insn_t in = w->insns[i];
        if (in.op == MS_OP_SLEEP_RAND) {
            uint64_t lo = (uint64_t)in.a * 50000000ull, hi = (uint64_t)in.b * 1000000000ull + in.imm;
            if (lo >= hi) return fail(err, MADSIM_E_WORKLOAD, "sleep_rand: cannot sample empty range");
            if (T->durs.size() / 4 > 255) return fail(err, MADSIM_E_WORKLOAD, "too many sleep_rand ops");
            uint32_t mode; uint64_t low, range, zone;
            uniform_duration_params(lo, hi, &mode, &low, &range, &zone);
            in.a = (uint8_t)(T->durs.size() / 4);
            T->durs.push_back(mode); T->durs.push_back(low); T->durs.push_back(range); T->durs.push_back(zone);
        }
        T->insns[4 * i] = (uint32_t)in.op | ((uint32_t)in.a << 8) | ((uint32_t)in.b << 16);
        T->insns[4 * i + 1] = in.imm;
        // Fused post-chain: the cheap ops that follow this one (an optional assert_eq!(val, ..) and an optional
        // loop-back / jump) are summarised in words 2-3, so completing an awaiting op steps straight to the next
        // awaiting op without dependent instruction fetches.  Pure acceleration: the ops keep their own records.
        //   word 3: bit0 has_assert | bit1 has_djnz | bit2 djnz reg | bit3 has_jmp | next_pc << 4 | target << 18
        uint32_t j = i + 1, flags = 0, aval = 0, target = 0;
        {
            if (j < w->n_insns && w->insns[j].op == MS_OP_ASSERT_VAL) { flags |= 1; aval = w->insns[j].imm; j++; }
            if (j < w->n_insns && w->insns[j].op == MS_OP_DJNZ) { flags |= 2 | ((w->insns[j].a & 1u) << 2); target = w->insns[j].b; j++; }
            else if (j < w->n_insns && w->insns[j].op == MS_OP_JMP) { flags |= 8; target = w->insns[j].b; j++; }
        }
        T->insns[4 * i + 2] = aval;
        T->insns[4 * i + 3] = flags | ((j & 0x3fffu) << 4) | ((target & 0x3fffu) << 18);
    }
    for (uint32_t i = 0; i < w->n_progs; i++) T->progs[i] = (uint32_t)w->progs[i].node | ((uint32_t)w->progs[i].flags << 8) | ((uint32_t)w->progs[i].entry << 16);
    // node table (sim_kernel.h KParams::nodes): flags words, restart rows, services — the layout make_geometry announced
    T->nodes.assign(w->n_nodes + 1, 0);
    bool rows = false;
    for (uint32_t i = 0; i <= w->n_nodes && w->nodes; i++) { T->nodes[i] = (uint32_t)w->nodes[i].flags; rows |= (w->nodes[i].flags & MADSIM_NODE_RESTART_MATCHING) != 0; }
    if (rows) {
        // bit c of row n: a panic with message code c restarts node n.  The caller's rows (madsim_workload_t.panic_match), or
        // equality on madsim_node_t.match[]; code 255 (failed asserts, unwraps: a message no pattern names) never matches.
        for (uint32_t i = 0; i <= w->n_nodes; i++) {
            uint32_t row[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (w->nodes[i].flags & MADSIM_NODE_RESTART_MATCHING) {
                if (w->panic_match) for (int k = 0; k < 8; k++) row[k] = w->panic_match[8 * i + k];
                else for (uint32_t k = 0; k < w->nodes[i].n_match && k < 2; k++) row[w->nodes[i].match[k] >> 5] |= 1u << (w->nodes[i].match[k] & 31);
                row[7] &= 0x7fffffffu;
            }
            T->nodes.insert(T->nodes.end(), row, row + 8);
        }
    }
    for (uint32_t k = 0; k < w->n_services; k++) {
        const madsim_service_t& sv = w->services[k];
        T->nodes.push_back((uint32_t)sv.vaddr | ((uint32_t)sv.n_servers << 8) | ((uint32_t)sv.servers[0] << 16) | ((uint32_t)sv.servers[1] << 24));
        T->nodes.push_back((uint32_t)sv.servers[2] | ((uint32_t)sv.servers[3] << 8) | ((uint32_t)sv.servers[4] << 16) | ((uint32_t)sv.servers[5] << 24));
    }
    return 0;
}

}  // namespace madsim_geo

#endif
