// k_channel.h — Reliable channel helpers (connect1 / accept1 / channel) and the link test as a function.
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_CHANNEL_H
#define MADSIM_K_CHANNEL_H

namespace madsim_k {

// ---- reliable channel (NetSim::connect1 / channel, net/mod.rs:337-430) — LIFE variants only ---------------------
// Network::try_send (network.rs:296-313) + test_link (:261-269) for a datagram from `src_node` to the address `addr`
// (node | kind << 8 | port << 16; `idx` = its socket-table entry when it has one, i.e. always for uniq_addr workloads).
// Returns 1 with the latency, the destination socket and the dst-was-loopback flag of `from` when a delivery must be
// scheduled, 0 when the message is dropped, -1 when the sender panics (`.ip.unwrap()` of an IP-less node, :309).
template <class K>
// `dst_hdr` receives the destination socket's header word (valid when the result is 1).
// `known_idx` / `known_hdr` (k_poll.h stage [A]): the header word of table entry known_idx, requested before the handlers — one wait for the wave
// instead of one in every handler that sends.
__device__ __forceinline__ int net_try_send(const Ctx& c, Lane& L, uint32_t src_node, uint32_t addr, uint32_t idx, uint64_t* latency, int* dst_sock, uint32_t* from_lb, uint32_t* dst_hdr,
                                            uint32_t known_idx = ~0u, uint32_t known_hdr = 0) {
    const KParams& P = c.P;
    // (one result variable, no early returns: every extra way out of an inlined body costs phi copies where it is used)
    int res = 0;
    // Global-state builds, plain addresses: the destination's table entry is known before the draws, so its header word is
    // requested here — with the clog words — and has arrived when the loss and latency draws are done, instead of costing a round
    // trip after them.  (Nothing is stored in between.)
    uint32_t hdr = 0;
    if (K::G && PLAIN_ADDR) {
        if (idx == known_idx) {
#ifdef MADSIM_EMU
            if (known_hdr != (uint32_t)SW(c, idx, 0)) OVF_SET(L, OVF_BUG);     // nothing was stored since the request
#endif
            hdr = known_hdr;
        } else hdr = SW(c, idx, 0);
    }
    int dn = (int)(addr & 0xff);                            // resolve_dest_node: plain node IPs resolve to their node
    if (!PLAIN_ADDR) dn = resolve_dest_node<K>(c, src_node, addr);          // < 0: dropped, no draw
    if (dn >= 0) {
        const uint32_t dst_node = (uint32_t)dn;
        bool clogged = false;
        // (extended builds mirror "some node is clogged" / "a link has been clogged" in the lane — Lane::loss_always bits 8, 9, kept by the
        //  clog ops of k_poll.h: while nothing is clogged a send loads none of the three mask words)
        const bool mir = K::LIFE && MADSIM_CLOG_MIRROR;
        if (P.has_clog && (!mir || (L.loss_always & 0x100u))) clogged = ((CLOGW(1) >> src_node) & 1) | ((CLOGW(0) >> dst_node) & 1);
        if (P.has_clog_link && (!mir || (L.loss_always & 0x200u))) clogged |= (CLOGW(2 + src_node) >> dst_node) & 1;
        if (!clogged && !gen_bool_pint<K>(c, L, L.loss_pint, K::LIFE ? (L.loss_always & 1u) : L.loss_always)) {
            L.msg_count++;
            *latency = sample_latency<K>(c, L);
            int ds;
            if (PLAIN_ADDR) { if (!K::G) hdr = SW(c, idx, 0); ds = (hdr & 1) ? (int)idx : -1; }     // (k_net.h find_bound)
            else {
                ds = find_exact<K>(c, dst_node, addr);              // sockets.get(&(dst, protocol))
                if (ds < 0) ds = find_exact<K>(c, dst_node, (addr & 0xffff0000u) | (MADSIM_ADDR_UNSPECIFIED << 8));   // .or_else(0.0.0.0:port)
                if (ds >= 0) hdr = SW(c, (uint32_t)ds, 0);
            }
            if (ds >= 0) {                                  // else: draws consumed, silently dropped
                *from_lb = 0;
                res = 1;
                if (!PLAIN_ADDR) {
                    *from_lb = ((addr >> 8) & 0xff) == MADSIM_ADDR_LOOPBACK;
                    if (!*from_lb && !node_has_ip(c, src_node)) res = -1;
                }
                *dst_sock = ds;
                *dst_hdr = hdr;
            }
        }
    }
    return res;
}

// the `test_link` closure of channel() (net/mod.rs:375-380): Some(now + latency), None (~0) or a panic of the caller
constexpr uint64_t CHAN_LINK_PANIC = ~0ull - 1;
template <class K>
__device__ __forceinline__ uint64_t chan_test_link(const Ctx& c, Lane& L, uint32_t cw, uint32_t dir) {
    // connect1 makes channel(node, dst) and channel(dst_node, src) (net/mod.rs:356-357): `dst` is the address connect1 was
    // GIVEN (entry d_ep — not the address of the socket that answered, which may be 0.0.0.0:port), `src` is what the
    // listener saw: the client's IP — 127.0.0.1 if dst is a loopback address — with the client Endpoint's port.
    const uint32_t c_ep = (cw >> 1) & 0x3f, d_ep = (cw >> 7) & 0x3f;
    const uint32_t cw_addr = SOCKW(c, c_ep), dw_addr = SOCKW(c, d_ep), dkind = (dw_addr >> 8) & 0xff;
    uint32_t src_node = cw_addr & 0xff, addr = dw_addr, idx = d_ep;
    if (dir) {
        if (!PLAIN_ADDR) src_node = dkind == MADSIM_ADDR_IP ? (dw_addr & 0xff) : src_node;   // dst_node (resolve_dest_node)
        else src_node = dw_addr & 0xff;
        addr = PLAIN_ADDR ? cw_addr : addr_of_from(c, c_ep | ((dkind == MADSIM_ADDR_LOOPBACK ? 1u : 0u) << 6));
        idx = c_ep;
    }
    uint64_t lat; int ds; uint32_t lb, dh;
    const int sent = net_try_send<K>(c, L, src_node, addr, idx, &lat, &ds, &lb, &dh);
    if (sent < 0) return CHAN_LINK_PANIC;                   // `.ip.unwrap()` inside try_send (network.rs:309)
    if (sent == 0) return ~0ull;
    return L.clock + lat;
}

// drop the raw (PayloadSender, PayloadReceiver) pair of one end of connection `id`
// (`cw`, `r` = the connection's header word and the parked-receiver word of this end's direction, when the caller holds them)
template <class K>
__device__ __forceinline__ void conn_drop_raw_with(const Ctx& c, Lane& L, uint32_t id, uint32_t side, uint32_t cw, uint32_t r) {
    if (cw & (1u << (13 + 2 * side))) {                       // my PayloadSender: last mpsc sender gone
        cw &= ~(1u << (13 + 2 * side));
        if (r & 1) { CONNW(id, 1 + side) = 0; CONNW(id, 0) = cw; wake<K>(c, L, (r >> 1) & 0xff, r >> 9); }   // parked receiver sees None
    }
    cw &= ~(1u << (14 + 2 * (1 - side)));                     // my PayloadReceiver
    CONNW(id, 1 + (1 - side)) = 0;
    if (!(cw & (0xfu << 13))) cw = 0;                          // all four handles gone: slot is free
    if (side) cw &= ~(0x7fu << 25);                            // (the listener-side guard reference goes with the handles)
    CONNW(id, 0) = cw;
}
template <class K>
__device__ __forceinline__ void conn_drop_raw(const Ctx& c, Lane& L, uint32_t id, uint32_t side) {
    const uint32_t cw = CONNW(id, 0);
    // (global-state builds: both words in one round trip; the LDS builds read the second one only when it is wanted)
    const uint32_t r = (Hoist<K>::CHAN || (cw & (1u << (13 + 2 * side)))) ? (uint32_t)CONNW(id, 1 + side) : 0u;
    conn_drop_raw_with<K>(c, L, id, side, cw, r);
}

// conn_tx / conn_rx of an Endpoint (endpoint.rs:18,307): up to MADSIM_ACCEPTQ connection ids waiting for accept1, as one
// 64-bit word — count in bits 0-3, 7-bit ids from bit 4 — in socket fields base (low half; its low 4 bits are the count, so
// "is anything queued" needs one load) and base + 2; field base + 1 is the parked acceptor.
constexpr uint32_t MADSIM_ACCEPTQ = 8;
template <class K> __device__ __forceinline__ uint64_t acceptq_load(const Ctx& c, uint32_t s) {
    const uint32_t base = 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs;
    return u64of(SW(c, s, base), SW(c, s, base + 2));
}
template <class K> __device__ __forceinline__ void acceptq_store(const Ctx& c, uint32_t s, uint64_t q) {
    const uint32_t base = 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs;
    SW(c, s, base) = (uint32_t)q; SW(c, s, base + 2) = (uint32_t)(q >> 32);
}

// the EndpointSocket is freed: the connections still queued in conn_tx go with it (nobody accepted them, so their
// server-side handles are the raw pair: no BindGuard clones)
template <class K>
__device__ __forceinline__ void sock_drop_acceptq(const Ctx& c, Lane& L, uint32_t s) {
    uint32_t base = 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs;
    uint64_t q = acceptq_load<K>(c, s);
    acceptq_store<K>(c, s, 0); SW(c, s, base + 1) = 0;
    uint32_t n = (uint32_t)q & 0xf;
    for (uint32_t i = 0; i < n; i++) conn_drop_raw<K>(c, L, (uint32_t)(q >> (4 + 7 * i)) & 0x7f, 1);
}

// Every Sender / Receiver holds a clone of its Endpoint's Arc<BindGuard> (endpoint.rs:181-190,203-210), so an address stays
// in the node's socket table until the Endpoint AND every connection end made from it are gone.  Socket header bits 25-31
// count the live (Sender, Receiver) pairs of a socket; an Endpoint dropped while that count is non-zero leaves the marker
// ~0 in the owner word instead of unbinding.  BindGuard::drop runs with the last owner and does nothing when the binder's
// NodeInfo is killed (net/mod.rs:483-493): reset_node has emptied the table — and the counts — by then.
// (`h` = the socket's header word, when the caller holds it)
template <class K>
__device__ __forceinline__ void guard_acquire_with(const Ctx& c, Lane& L, uint32_t s, uint32_t h) {
    if ((h >> 25) == 0x7fu) { OVF_SET(L, OVF_MODEL); return; }          // (a seven-bit count: 127 connection ends per socket — no limit grows it: the model's edge, the oracle says the same)
    SW(c, s, 0) = h + (1u << 25);
}
template <class K>
__device__ __forceinline__ void guard_acquire(const Ctx& c, Lane& L, uint32_t s) { guard_acquire_with<K>(c, L, s, SW(c, s, 0)); }
template <class K>
__device__ __forceinline__ void guard_release(const Ctx& c, Lane& L, uint32_t s, bool node_killed) {
    if (node_killed) return;
    uint32_t h = SW(c, s, 0);
    const uint32_t own_pre = Hoist<K>::CHAN ? (uint32_t)SW(c, s, 1) : 0u;      // (global-state builds: the owner word with the header)
    if (!(h >> 25)) return;
    h -= 1u << 25;
    if (!(h >> 25) && (h & 1) && (Hoist<K>::CHAN ? own_pre : (uint32_t)SW(c, s, 1)) == ~0u) {       // the last owner: Network::close, and the socket dies
        SW(c, s, 0) = h & ~1u;
        if (SW(c, s, 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs) & 0xf) sock_drop_acceptq<K>(c, L, s);
    } else {
        SW(c, s, 0) = h;
    }
}

// drop(tx); drop(rx) of one end of connection `id`, by a task whose NodeInfo is killed or not
template <class K>
__device__ __forceinline__ void conn_drop_handles(const Ctx& c, Lane& L, uint32_t id, uint32_t side, bool node_killed) {
    const uint32_t cw = CONNW(id, 0);
    const uint32_t r = (Hoist<K>::CHAN || (cw & (1u << (13 + 2 * side)))) ? (uint32_t)CONNW(id, 1 + side) : 0u;
    const bool held = side ? (cw >> 31) != 0 : (cw & (1u << 13)) != 0;       // accepted / the client's handles exist
    const uint32_t gs = side ? (cw >> 25) & 0x3f : (cw >> 1) & 0x3f;
    conn_drop_raw_with<K>(c, L, id, side, cw, r);
    if (held) guard_release<K>(c, L, gs, node_killed);
}

// drop(Endpoint) of socket s by its owner: conn_rx goes (later connections are dropped on arrival, endpoint.rs:320-328) and
// the Arc<BindGuard> loses one owner
template <class K>
__device__ __forceinline__ void endpoint_drop(const Ctx& c, Lane& L, uint32_t s, bool node_killed) {
    const uint32_t h = SW(c, s, 0);
    const bool chan = K::FC && c.P.uses_chan;
    const uint32_t base = 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs;
    if (chan) SW(c, s, base + 1) = 0;                          // the parked acceptor was the owner
    bool bound = h & 1;
    // BindGuard::drop -> Network::close, unless connections hold clones of the guard or the binder's NodeInfo is killed
    // (after Handle::kill reset_node has emptied the table already; after an init task's exit the address just stays)
    if (bound && !node_killed && !(chan && (h >> 25))) { SW(c, s, 0) = h & ~1u; bound = false; }
    if (bound) { if (K::LIFE) SW(c, s, 1) = ~0u; }             // the table keeps the EndpointSocket (and its queue) alive
    else if (chan && (SW(c, s, base) & 0xf)) sock_drop_acceptq<K>(c, L, s);
}

}  // namespace madsim_k

#endif
