// k_net.h — async-task wake rules, socket lookup, Mailbox::deliver, Timer::expire.
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_NET_H
#define MADSIM_K_NET_H

namespace madsim_k {

// ---- async-task wake / schedule [DEP A.7] -------------------------------------------------------
template <class K>
__device__ __forceinline__ void ready_push(const Ctx& c, Lane& L, uint32_t slot) {
    if (K::RQ) L.rq |= (uint64_t)slot << (8 * L.ready_len);
    else rq_set<K>(c, L.ready_len, slot);
    L.ready_len++;
}

// (`f` = the task's flag word when the caller has loaded it already: timer_expire's prefetch)
template <class K>
__device__ __forceinline__ void wake_with(const Ctx& c, Lane& L, uint32_t slot, uint32_t gen, const uint32_t f) {
    // not COMPLETED | CLOSED (a stale waker), not SCHEDULED already
    if ((f & (TF_ALIVE | TF_SCHED)) == TF_ALIVE && ((f >> 8) & 0xffff) == gen) {
        TWORD(c, slot, 0, 0) = f | TF_SCHED;
        if (!(f & TF_RUN)) ready_push<K>(c, L, slot);               // RUNNING: run() re-queues after the poll
    }
}
template <class K>
__device__ __forceinline__ void wake(const Ctx& c, Lane& L, uint32_t slot, uint32_t gen) {
    wake_with<K>(c, L, slot, gen, TWORD(c, slot, 0, 0));
}

// ---- Network -----------------------------------------------------------------------------------
// A SocketAddr as one word, the form of the socket table: node | kind << 8 | port << 16 (kind = MADSIM_ADDR_*).
__device__ __forceinline__ uint32_t addr_of_from(const Ctx& c, uint32_t from) {   // the address a receiver was shown
    uint32_t w = SOCKW(c, from & 0x3f) & 0xffff00ffu;                             // sender's real IP and socket port,
    return w | (((from & 0x40) ? MADSIM_ADDR_LOOPBACK : MADSIM_ADDR_IP) << 8);    // or 127.0.0.1 (network.rs:307-311)
}
__device__ __forceinline__ bool addr_eq(uint32_t x, uint32_t y) {                 // SocketAddr equality
    const uint32_t kind = (x >> 8) & 0xff;                                         // (0.0.0.0 and 127.0.0.1 are the same address on every node)
    return ((x ^ y) & 0xffffff00u) == 0 && ((kind != MADSIM_ADDR_IP && kind != MADSIM_ADDR_VIRTUAL) || ((x ^ y) & 0xffu) == 0);
}
__device__ __forceinline__ bool node_has_ip(const Ctx& c, uint32_t node) { return !(NODET(c, node) & MADSIM_NODE_NO_IP); }

// node.sockets.get(&(addr, protocol)) on node `on` (network.rs keys a node's sockets by the address they were bound to)
template <class K>
__device__ __forceinline__ int find_exact(const Ctx& c, uint32_t on, uint32_t addr) {
    const uint32_t key = (addr & 0xffffff00u) | on;
    if (((addr >> 8) & 0xff) == MADSIM_ADDR_IP && (addr & 0xff) != on) return -1;
    for (uint32_t i = 0; i < c.P.n_socks; i++)
        if (SOCKW(c, i) == key && (SW(c, i, 0) & 1)) return (int)i;
    return -1;
}
// Network::try_send's socket lookup (network.rs:304-306) for workloads whose addresses are all distinct node IPs
// (KParams.uniq_addr: the common case): the bound socket at table entry `addr`, if any.
template <class K>
__device__ __forceinline__ int find_bound(const Ctx& c, uint32_t addr) {
    return (SW(c, addr, 0) & 1) ? (int)addr : -1;
}
// Network::resolve_dest_node (network.rs:272-290): the node a datagram for `addr` goes to, or -1 (dropped, no draws)
template <class K>
__device__ __forceinline__ int resolve_dest_node(const Ctx& c, uint32_t node, uint32_t addr) {
    const uint32_t kind = (addr >> 8) & 0xff, an = addr & 0xff;
    if (kind == MADSIM_ADDR_LOOPBACK || find_exact<K>(c, node, addr) >= 0) return (int)node;
    if (!node_has_ip(c, node)) return -1;                                          // "ip not set"
    if (kind == MADSIM_ADDR_IP && an >= 1 && an <= c.P.n_nodes && node_has_ip(c, an)) return (int)an;   // addr_to_node
    return -1;                                                                     // "destination not found"
}

// `if let Some(addr) = ipvs.get_server(dst) { dst = addr }` of NetSim::send / connect1 (net/mod.rs:312-317,345-350) with
// IpVirtualServer::get_server (net/ipvs.rs:88-105): the service whose address equals `addr`; no servers -> no rewrite;
// `if *i >= len { *i = 0 }; server = servers[*i]; *i += 1`.  `idx` = the table entry `addr` came from, rewritten with it.
template <class K>
__device__ __forceinline__ void ipvs_rewrite(const Ctx& c, uint32_t& idx, uint32_t& addr) {
    if (!K::FA || !c.P.n_services) return;
    bool found = false;
    for (uint32_t k = 0; k < c.P.n_services && !found; k++) {
        const uint32_t w0 = SMEM[c.nodet0 + c.P.svc_off + 2 * k], w1 = SMEM[c.nodet0 + c.P.svc_off + 2 * k + 1];
        if (addr_eq(SOCKW(c, w0 & 0xff), addr)) {
            found = true;
            if (c.P.ipvs_dyn) {                 // the service's state is the seed's own (MS_OP_IPVS, k_poll.h): {servers[0..3]}, {servers[4..5], n, rr, present}
                const uint32_t d0 = IPVSW(2 * k), d1 = IPVSW(2 * k + 1), n = (d1 >> 16) & 0xf;
                if ((d1 >> 24) & 1u && n) {
                    uint32_t i = (d1 >> 20) & 0xf;
                    if (i >= n) i = 0;
                    IPVSW(2 * k + 1) = (d1 & ~(0xfu << 20)) | ((i + 1) << 20);
                    const uint32_t srv = i < 4 ? (d0 >> (8 * i)) & 0xff : (d1 >> (8 * (i - 4))) & 0xff;
                    idx = srv; addr = SOCKW(c, srv);
                }
                continue;
            }
            const uint32_t n = (w0 >> 8) & 0x7;      // (bit 7: declared absent — no servers either)
            if (n) {
                uint32_t i = IPVSW(k);
                if (i >= n) i = 0;
                IPVSW(k) = i + 1;
                const uint32_t lo = w0 >> 16, srv = i < 2 ? (lo >> (8 * i)) & 0xff : (w1 >> (8 * (i - 2))) & 0xff;
                idx = srv; addr = SOCKW(c, srv);
            }
        }
    }
}

// Mailbox::deliver (endpoint.rs:331-351)
// `pf` (global-state builds): the socket's header word and first registration, loaded by timer_expire before the pop's sift
struct DeliverPrefetch { uint32_t hdr, reg0; };
template <class K>
__device__ __forceinline__ void mailbox_deliver(const Ctx& c, Lane& L, uint32_t meta, uint32_t val, const DeliverPrefetch* pf = nullptr) {
    uint32_t s = meta & 0x3f, from = (meta >> 6) & 0x7f, tag = (meta >> 13) & 0xff, sgen = (meta >> 21) & 0xff;
    if (!K::LIFE) {                                        // base ops: the event names its send_to / reply instruction
        const uint4 in = INSN(c, (meta >> 6) & 0xfff);
        from = (in.x >> 8) & 0x3f; tag = in.x >> 24; val = in.y;      // (base-op builds: plain addresses, never a loopback flag)
    }
    const uint32_t h = pf ? pf->hdr : (uint32_t)SW(c, s, 0);
    bool live = (h & 1) && ((h >> 1) & 0xff) == sgen;      // else: that Endpoint object is gone
    // The Endpoint object is gone (its address is still held: by connections made from it, or because its node was killed before the
    // BindGuard dropped — `if self.node.is_killed() { return }`, net/mod.rs:483-493 — and TaskHandle::restart resets no sockets): the
    // EndpointSocket stays in the table and its mailbox still takes a message for a receive that some OTHER holder of the Endpoint
    // registered; with nobody left to register one, a message is dropped instead of queued for ever.
    if (live) {
        uint32_t nreg = (h >> 9) & 0xff, nmsg = HDR_NMSG(h);
        // Typed RPC (net/rpc.rs): a response (tag 0xff) is addressed to one pending receive — the word of its registration,
        // tag | slot | rxseq | gen, rides in the payload's upper 24 bits — where the reference matches a random u64 tag.
        const bool rpc = K::FR && c.P.uses_rpc;
        const bool rsp = rpc && tag == 0xff;
        uint32_t i = 0;
        bool taken = false;                                // (one loop exit, no early returns: see k_main.h)
        // Global-state builds: `known` = the value of entry i is in `r_known` — the prefetched entry 0 before any removal, and after a
        // swap_remove the entry just moved into position i.  A dead registration (a timed-out receive: the election loop's mailboxes
        // are full of them) then costs ONE round trip — its task's words and the entry to move, requested together — not three
        // (entry, entry to move, task words, each waiting for the one before).
        bool known = pf != nullptr;
        uint32_t r_known = pf ? pf->reg0 : 0u;
        while (i < nreg && !taken) {
            REG(24);
            uint32_t r = known ? r_known : (uint32_t)SW(c, s, 2 + i);
            known = false;
            if ((r & 0xff) == tag && (!rsp || (r >> 8) == (val >> 8))) {
                nreg--;
                uint32_t slot = (r >> 8) & 0xff, rxseq = (r >> 16) & 0xff, g8 = r >> 24;
                // swap_remove: the three loads first, then the store.  (Global-state builds: when the match IS the last registration — the
                // usual mailbox holds one — nothing moves: no load of the last entry, no store over a position that is no longer part of
                // the list.)
                const bool last = K::G && i == nreg;
                const uint32_t moved = last ? r : (uint32_t)SW(c, s, 2 + nreg);
                uint4 u0 = TU(c, slot, 0);
                uint32_t link = TWORD(c, slot, 1, 0);
                if (!last) SW(c, s, 2 + i) = moved;
                if (K::G) { known = true; r_known = moved; }
                if ((u0.x & TF_ALIVE) && ((u0.x >> 8) & 0xff) == g8 && (link & 0xff) == rxseq && !(u0.x & TF_INBOX)) {
                    // oneshot::Sender::send Ok -> value stored, receiver task woken [DEP tokio oneshot]
                    bool sched = u0.x & TF_SCHED;
                    u0.x |= TF_INBOX | TF_SCHED;
                    u0.y = (u0.y & 0x00ffffffu) | (from << 24);
                    u0.w = val;
                    if (rpc && tag >= MADSIM_TAG_RPC_FIRST) {                  // 8-bit code; a request also carries its rsp_tag
                        u0.w = val & 0xff;
                        if (!rsp) TWORD(c, slot, c.P.rpc_unit, 1) = val >> 8;  // staged with the oneshot value
                    }
                    TU(c, slot, 0) = u0;
                    if (!sched && !(u0.x & TF_RUN)) ready_push<K>(c, L, slot);
                    taken = true;
                }
            } else {
                i++;
            }
        }
        if (taken) {
            SW(c, s, 0) = (h & ~(0xffu << 9)) | (nreg << 9);
        } else if (K::LIFE && SW(c, s, 1) == ~0u) {              // (the owner word is read only when nobody took the message)
            SW(c, s, 0) = (h & ~(0xffu << 9)) | (nreg << 9);      // (dead registrations swept on the way stay swept)
        } else if (nmsg >= c.P.mbox_msgs) {
            // (short of the ceiling of 255 queued messages a larger mbox_msgs lifts it; AT the ceiling the 256th leaves the model)
            OVF_SET(L, c.P.mbox_msgs >= MADSIM_MAX_MBOX_MSGS ? OVF_MODEL : OVF_CAP);
        } else {
            if (rsp) tag = 0xfe;                           // nobody holds that rsp_tag any more: it can never be received
            SW(c, s, 2 + c.P.mbox_regs + 2 * nmsg) = tag | (from << 8);
            SW(c, s, 2 + c.P.mbox_regs + 2 * nmsg + 1) = val;
            nmsg++;
            SW(c, s, 0) = (HDR_SET_NMSG(h, nmsg) & ~(0xffu << 9)) | (nreg << 9);
        }
    }
}

template <class K> __device__ __forceinline__ void node_restart(const Ctx& c, Lane& L, uint32_t node);

// Timer::expire [DEP A.5]: fire every entry with deadline <= now
template <class K>
__device__ __forceinline__ void timer_expire(const Ctx& c, Lane& L, uint64_t now) {
#ifdef MADSIM_EMU
    // Global-state builds queue a round's Timer::add calls in the lane (k_timer.h timer_schedule): whoever reads the heap — top_dl, a
    // pop — must come after timer_flush.  Checked in the host-compiled kernel on every call (the emulation suite asserts that no
    // seed ends MADSIM_INTERNAL); a new op that fires timers inside a round without flushing first shows up there.
    if (K::G && L.pq_n) OVF_SET(L, OVF_BUG);
#endif
    // Copies of a wake-up (below) — form 2: popped by this loop itself (builds with a spill region: the topology's); form 1: a loop of their own
    // behind the wake-up (the other global-state builds: the KV's channel build, two spilled registers already, read 0.5 % slower with form 2's three
    // live values).  last_dl / last_meta = the wake-up fired last in this call, which form 2 recognises its copies by.
    constexpr bool COPIES2 = MADSIM_FIRE_COPIES == 2 && K::G && !K::DEDUP && K::SPILL;
    constexpr bool COPIES1 = MADSIM_FIRE_COPIES != 0 && K::G && !K::DEDUP && !COPIES2;
    uint64_t last_dl = ~0ull;
    uint32_t last_meta = 0;
    while (L.top_dl <= now) {
        // Global-state builds: the callback's first loads (the woken task's flag word; the destination socket's header and
        // first registration) depend only on the ROOT entry, which sits in LDS — issue them before the pop, whose sift-down
        // walks the spilled heap levels one dependent round trip at a time, so they arrive with its first level instead of
        // costing a round trip of their own afterwards.  The pop touches neither task nor socket state.
        uint32_t pf_flags = 0;
        DeliverPrefetch pf = {0, 0};
        uint2 pf_rec = make_uint2(0, 0);                  // narrow-heap builds: the root delivery's pool record {full event word, payload}
        bool copy = false;
        if (K::G) {
            const uint32_t rz = heap_root_meta<K>(c), rk = rz >> EV_SHIFT;
            if (COPIES2) copy = L.top_dl == last_dl && rz == last_meta;
            if (copy) { }
            else if (rk == EV_WAKE) pf_flags = TWORD(c, rz & 0xff, 0, 0);
            else if (rk == EV_DELIVER) {
                pf.hdr = SW(c, rz & 0x3f, 0); pf.reg0 = SW(c, rz & 0x3f, 2);
                if (K::NH) pf_rec = buf_load64(c.gs, pool_addr(c, (rz >> 6) & 0x7fffu));
            }
        }
        // DEDUP builds: the bucket that may hold repeats of the root wake-up (k_timer.h dedup_note) — like the loads above it
        // depends on the root entry alone
        const bool dd = K::DEDUP && c.P.dedup_n && !L.exact;
        uint32_t dd_at = 0, dd_ix = 0;
        uint4 dd_u = make_uint4(0, 0, 0, 0);
        if (K::DEDUP && dd) {
            const uint32_t rz = heap_root_meta<K>(c);
            if ((rz >> EV_SHIFT) == EV_WAKE) {
                dd_ix = dedup_index(c, L.top_dl, rz); dd_at = dedup_at(c, dd_ix);
                if (!MADSIM_DEDUP_OCC || ((L.dd_occ >> dd_ix) & 1ull)) dd_u = gs_load128(c.gs, gs_addr_unit(c, dd_at));     // (an empty bucket: nothing to load)
            }
        }
        uint4 e = timer_pop<K>(c, L);
        L.steps++;
        uint32_t kind = e.z >> EV_SHIFT;
        const uint32_t popped_meta = e.z;                 // (as it sat in the heap: what a tie is recognised by)
        if (K::NH && kind == EV_DELIVER) { pool_free<K>(c, (e.z >> 6) & 0x7fffu); e.z = pf_rec.x; e.w = pf_rec.y; }
        if (K::DEDUP && dd) {
            // the repeats of this wake-up fire with it: a step each, their task is SCHEDULED by the first already
            if (dd_u.w != 0 && dd_u.x == e.x && dd_u.y == e.y && dd_u.z == e.z) { L.steps += dd_u.w; gs_store32(c.gs, gs_addr_uword(c, dd_at + 12u), 0); if (MADSIM_DEDUP_OCC) L.dd_occ &= ~(1ull << dd_ix); }
            // two DIFFERENT events with one deadline: which fires first is a matter of the heap's shape (restart the seed exactly)
            if (L.heap_len > 0 && L.top_dl == ev_deadline(e)) {
                if (K::NH) { if (heap_root_meta<K>(c) != popped_meta) L.hazard = 1; }     // (two deliveries never share a record: always different)
                else { const uint4 r = heap_lds_get<K>(c, 0); if (r.z != e.z || r.w != e.w) L.hazard = 1; }
            }
        }
        // Form 2: a copy of the wake-up fired last is popped by THIS loop — a step, no callback — in the same trips as the other
        // lanes' pops (form 1 popped them in a loop of its own behind the wake-up: 2.2 wave trips a pass with five lanes in them,
        // tools/mem_site_model.py topo).  The same entries leave the heap in the same order.
        if (COPIES2 && copy) { }
        else if (kind == EV_WAKE) {                                                         // time/sleep.rs:52
            REG(22);
            if (K::G) wake_with<K>(c, L, e.z & 0xff, (e.z >> 8) & 0xffff, pf_flags);
            else wake<K>(c, L, e.z & 0xff, (e.z >> 8) & 0xffff);
            if (COPIES2) { last_dl = ev_deadline(e); last_meta = popped_meta; }
            // Copies of this wake-up (the re-registrations of one pending Sleep: same deadline, same waker) are next in line, and firing them
            // changes nothing: the first made its task SCHEDULED — or found it gone, or of another generation — and so does every copy.
            // They are popped here, a step each (Timer::expire counts every entry), without the flag-word load and the store of a
            // callback.  Nothing else can sit between them: the heap hands out equal deadlines in ITS order, and only an entry equal in
            // deadline AND event word is skipped.
            if (COPIES1) {
                while (L.heap_len > 0 && L.top_dl == ev_deadline(e) && heap_root_meta<K>(c) == popped_meta) { (void)timer_pop<K>(c, L); L.steps++; }
            }
        }
        else if (kind == EV_DELIVER) { REG(23); mailbox_deliver<K>(c, L, e.z, e.w, K::G ? &pf : nullptr); last_dl = ~0ull; }      // net/mod.rs:323-330
        else if (K::FN && kind == EV_RESTART) { node_restart<K>(c, L, e.z & 0xff); last_dl = ~0ull; }   // task/mod.rs:313
        else last_dl = ~0ull;
    }
}

}  // namespace madsim_k

#endif
