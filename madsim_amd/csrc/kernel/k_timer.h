// k_timer.h — Timer: the BinaryHeap of naive-timer, LDS-resident with an HBM spill region.
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_TIMER_H
#define MADSIM_K_TIMER_H

namespace madsim_k {

// ---- Timer = BinaryHeap<Event>, reversed Ord on deadline [DEP naive-timer 0.2 + alloc BinaryHeap] --
// entry: x = deadline lo, y = deadline hi, z = meta, w = payload value
__device__ __forceinline__ uint64_t ev_deadline(const uint4& e) { return u64of(e.x, e.y); }

// Entries [0, heap_lds) live in LDS; entries beyond spill to HBM as [slot][global lane] (coalesced
// across the wave).  Variants without a spill region drop the HBM path.  The LDS load is issued
// unconditionally (clamped index) and the HBM value selected afterwards, so the two address spaces
// never merge into a flat_* access.
// (Tried in round 3: [sibling pair][lane][2] — both children of a sift-down level in one 32-byte granule, one sector per
// lane and level instead of two.  A/B on one box: election loop +2 %, topology 0, timer storm -2 %: not kept.)
__device__ __forceinline__ uint4 spill_load(const Ctx& c, uint32_t slot) {
    return buf_load128(c.spill, slot * c.P.total_lanes * 16u + c.spill_off);
}
__device__ __forceinline__ void spill_store(const Ctx& c, uint32_t slot, const uint4& e) {
    buf_store128(c.spill, slot * c.P.total_lanes * 16u + c.spill_off, e);
}

// ---- Variant::NH: 8-byte entries {low deadline word, event word} in LDS and in the spill region ---------------------------------
// The full deadline is clock + sign-extended (low word - low word of the clock): exact while every live deadline lies within 2^31 ns
// of the clock (timer_add checks each push; the compact base-op layout below does the same with a host-side proof).  The comparisons
// of push / pop run on these reconstructed 64-bit values, i.e. on the same numbers as the 16-byte layout.
// Event word: EV_WAKE / EV_RESTART / EV_NOP as everywhere; EV_DELIVER: kind << 29 | socket gen << 21 | pool record << 6 | destination
// socket — what the pop's prefetch needs (socket, record); tag, sender and payload wait in the record {full event word, payload}.
constexpr uint64_t NH_HORIZON = (1ull << 31) - (1ull << 24);
__device__ __forceinline__ uint4 nh_expand(const Lane& L, const uint2& e) {
    const uint64_t d = L.clock + (uint64_t)(int64_t)(int32_t)(e.x - (uint32_t)L.clock);
    return make_uint4((uint32_t)d, (uint32_t)(d >> 32), e.y, 0);
}
// Layout: position i as one 8-byte row element [i][lane] in LDS, [i - heap_lds][global lane] in the spill region.  (Round 6 also built SIBLING
// PAIRS — positions 2q + 1 and 2q + 2 in one 16-byte unit, one access per sift-down level — bit-exact, measured on one MI355X: topology
// 5.17 against 5.36 G steps/s for the rows (-3.6 %), election loop +1 % (noise): the single-entry stores of a sift, which dominate, coalesce
// half as well at a 16-byte lane stride.  the layout is in the tree at commit ae53d2e; profiles/r6_experiments.md.)
template <class K>
__device__ __forceinline__ uint2 nh_load(const Ctx& c, uint32_t i) {
    const uint32_t cap = c.P.heap_lds;
    uint2 v;
    if (!K::SPILL || i < cap) v = LDS64(c.heap0 + (i << LWSH<K>(c)));
    else v = buf_load64(c.spill, (i - cap) * c.P.total_lanes * 8u + c.spill_off);
    return v;
}
template <class K>
__device__ __forceinline__ void nh_store(const Ctx& c, uint32_t i, const uint2& v) {
    if (!K::SPILL || i < c.P.heap_lds) LDS64(c.heap0 + (i << LWSH<K>(c))) = v;
    else buf_store64(c.spill, (i - c.P.heap_lds) * c.P.total_lanes * 8u + c.spill_off, v);
}
// the delivery-record pool: record r of this lane at [r][global lane], 8 bytes, behind the planes of the state buffer
__device__ __forceinline__ uint32_t pool_addr(const Ctx& c, uint32_t r) { return __umul24(c.P.pool_off + r * 8u, c.P.total_lanes) + c.gs_lane * 8u; }
template <class K>
__device__ __forceinline__ uint32_t pool_alloc(const Ctx& c) {          // lowest free record, or ~0
    uint32_t r = ~0u;
    for (uint32_t w = 0; w < c.P.pool_n / 32 && r == ~0u; w++) {
        const uint32_t m = PMASK(w);
        if (~m) { const uint32_t b = (uint32_t)__builtin_ctz(~m); PMASK(w) = m | (1u << b); r = w * 32 + b; }
    }
    return r;
}
template <class K>
__device__ __forceinline__ void pool_free(const Ctx& c, uint32_t r) { PMASK(r >> 5) &= ~(1u << (r & 31)); }
// the event word of the root entry (its kind and operands steer timer_expire's prefetch)
template <class K> __device__ __forceinline__ uint32_t heap_root_meta(const Ctx& c);

// LDS entry i of this lane.  Extended builds: one 16-byte unit {deadline, meta, payload}.  Base-op builds: 12 bytes — the
// deadline in an 8-byte array, meta in a 4-byte one, no payload word: their only events with a payload are datagram
// deliveries, whose tag / sender / payload are fields of the sending instruction, so meta names that instruction instead
// (ev_deliver_meta, mailbox_deliver).
template <class K>
__device__ __forceinline__ uint4 heap_lds_get(const Ctx& c, uint32_t i) {
    if (K::LIFE) return LDS128(c.heap0 + (i << LWSH<K>(c)));
    uint2 d = LDS64(c.heap0 + (i << LWSH<K>(c)));
    return make_uint4(d.x, d.y, SMEM[c.heapm0 + (i << LWSH<K>(c))], 0);
}
template <class K>
__device__ __forceinline__ void heap_lds_set(const Ctx& c, uint32_t i, const uint4& e) {
    if (K::LIFE) { LDS128(c.heap0 + (i << LWSH<K>(c))) = e; return; }
    LDS64(c.heap0 + (i << LWSH<K>(c))) = make_uint2(e.x, e.y);
    SMEM[c.heapm0 + (i << LWSH<K>(c))] = e.z;
}
// Compact base-op builds (sim_kernel.h MADSIM_FEAT_COMPACT): entry 0 lives in registers (Lane::top_dl, top_meta), entry
// i >= 1 is 8 bytes in LDS — the low 32 bits of its deadline and its meta word.  The host admits the layout only when every
// live deadline lies within 2^31 ns of the clock (geometry.h: the workload's longest sleep), so clock + sign-extended
// (low word - low word of the clock) IS the deadline: the comparisons below run on the same 64-bit values as everywhere else.
template <class K> __device__ __forceinline__ uint32_t heap_root_meta(const Ctx& c) {
    if (K::NH) return LDS64(c.heap0).y;
    return heap_lds_get<K>(c, 0).z;
}
template <class K>
__device__ __forceinline__ uint4 heap_get(const Ctx& c, const Lane& L, uint32_t i) {
    if (K::NH) return nh_expand(L, nh_load<K>(c, i));
    if (K::CMP) {
        const uint2 e = LDS64(c.heap0 + ((i ? i - 1 : 0) << LWSH<K>(c)));
        const uint64_t d = L.clock + (uint64_t)(int64_t)(int32_t)(e.x - (uint32_t)L.clock);
        return i == 0 ? make_uint4((uint32_t)L.top_dl, (uint32_t)(L.top_dl >> 32), L.top_meta, 0) : make_uint4((uint32_t)d, (uint32_t)(d >> 32), e.y, 0);
    }
    if (!K::SPILL) return heap_lds_get<K>(c, i);
    // (a branch, not an unconditional LDS read of a clamped index overwritten by the spilled value: that form made every spilled
    // access wait for an LDS round trip first — the compiler overwrites the LDS result's registers under the other lanes' mask and
    // must see the LDS read complete before it may issue the buffer load.  Both sides are explicit address spaces — ds_* and
    // buffer_* — so nothing can merge them into flat_* accesses.)
    const uint32_t cap = c.P.heap_lds;
    uint4 v;
    if (i < cap) v = heap_lds_get<K>(c, i); else v = spill_load(c, i - cap);
    return v;
}
// Two entries lo < hi at once (the children of a sift-down level; parent and grandparent of a sift-up trip, hi = the deeper one):
// when both are spilled — every level below the LDS part of the heap — the two buffer loads go out back to back with no LDS
// access between them.
template <class K>
__device__ __forceinline__ void heap_get2(const Ctx& c, const Lane& L, uint32_t lo, uint32_t hi, uint4& vlo, uint4& vhi) {
    if (K::NH) {
        const uint32_t cap = c.P.heap_lds;
        uint2 a, b;
        if (K::SPILL && lo >= cap) {       // both spilled: the two loads back to back
            a = buf_load64(c.spill, (lo - cap) * c.P.total_lanes * 8u + c.spill_off); b = buf_load64(c.spill, (hi - cap) * c.P.total_lanes * 8u + c.spill_off);
        } else { a = LDS64(c.heap0 + (lo << LWSH<K>(c))); b = nh_load<K>(c, hi); }
        vlo = nh_expand(L, a); vhi = nh_expand(L, b);
        return;
    }
    if (K::SPILL && !K::CMP) {
        const uint32_t cap = c.P.heap_lds;
        if (lo >= cap) { vlo = spill_load(c, lo - cap); vhi = spill_load(c, hi - cap); }
        else { vlo = heap_lds_get<K>(c, lo); vhi = heap_get<K>(c, L, hi); }
        return;
    }
    vlo = heap_get<K>(c, L, lo); vhi = heap_get<K>(c, L, hi);
}
template <class K>
__device__ __forceinline__ void heap_set(const Ctx& c, Lane& L, uint32_t i, const uint4& e) {
    if (K::NH) { nh_store<K>(c, i, make_uint2(e.x, e.z)); return; }
    if (K::CMP) {
        if (i == 0) { L.top_dl = u64of(e.x, e.y); L.top_meta = e.z; }
        else LDS64(c.heap0 + ((i - 1) << LWSH<K>(c))) = make_uint2(e.x, e.z);
        return;
    }
    if (!K::SPILL || i < c.P.heap_lds) heap_lds_set<K>(c, i, e);
    else spill_store(c, i - c.P.heap_lds, e);
}
// The children child, child + 1 of a level that lies in LDS, and a store to an LDS-resident slot: ds_* accesses only, nothing that waits on
// the global-memory counter (timer_pop's walk through the LDS levels runs beside a spill-region load).
template <class K>
__device__ __forceinline__ void heap_lds_get2(const Ctx& c, const Lane& L, uint32_t child, uint4& vlo, uint4& vhi) {
    if (K::NH) {
        const uint2 a = LDS64(c.heap0 + (child << LWSH<K>(c))), b = LDS64(c.heap0 + ((child + 1) << LWSH<K>(c)));
        vlo = nh_expand(L, a); vhi = nh_expand(L, b);
    } else { vlo = heap_lds_get<K>(c, child); vhi = heap_lds_get<K>(c, child + 1); }
}
template <class K>
__device__ __forceinline__ void heap_set_lds(const Ctx& c, Lane& L, uint32_t i, const uint4& e) {
    if (K::NH) LDS64(c.heap0 + (i << LWSH<K>(c))) = make_uint2(e.x, e.z);
    else heap_lds_set<K>(c, i, e);
}
// meta + payload of a datagram delivery event (net/mod.rs:323-330): destination socket `ds` of incarnation `sgen`.
template <class K>
__device__ __forceinline__ uint2 ev_deliver_meta(uint32_t sgen, uint32_t tag, uint32_t from, uint32_t ds, uint32_t val, uint32_t pc) {
    if (K::LIFE) return make_uint2((EV_DELIVER << EV_SHIFT) | (sgen << 21) | (tag << 13) | (from << 6) | ds, val);
    return make_uint2((EV_DELIVER << EV_SHIFT) | (sgen << 21) | (pc << 6) | ds, 0);     // base ops: tag, from, payload = fields of insn pc
}

// BinaryHeap::sift_up(0, pos) with `hole` as the moving element; keeps the root mirror current.
// Builds with a spill region walk two levels per trip: parent and grandparent are loaded together (their indices depend on
// `pos` alone), so a sift through the spilled levels costs one global round trip per TWO levels; the comparisons and stores
// are those of the one-level loop, in the same order.
// `pre` (k_poll.h: MADSIM_PUSH_PREFETCH): the entry of heap slot pre.idx, requested when the poll round began — the parent of the slot the round's
// first push starts from, a spilled one whose own parent lies in LDS — so that push compares at once instead of waiting a round trip.
struct HeapPre { uint32_t idx; uint2 e; };
// (the every-class builds: two waves per SIMD, registers to spare.  The single-class ones run three at 168 registers — the election loop's spilled
//  three with this — and their heaps are short: a push's parent is an LDS entry there.)
template <class K> struct PushPrefetch { static constexpr bool ON = MADSIM_PUSH_PREFETCH && K::G && K::NH && K::SPILL && (K::FEAT & (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR)) == (MADSIM_FEAT_ALL & ~MADSIM_FEAT_ADDR); };
template <class K>
__device__ __forceinline__ void heap_sift_up(const Ctx& c, Lane& L, uint32_t pos, const uint4& hole, HeapPre pre = HeapPre{~0u, make_uint2(0, 0)}) {
    uint64_t hd = ev_deadline(hole);
    bool up = pos > 0;
    while (up) {                                       // (one exit: see k_main.h)
        REG(11);
        const uint32_t parent = (pos - 1) >> 1;
        if (K::SPILL || K::G) {          // (global-state builds without a spill region too: two LDS levels per round trip)
            const uint32_t gp = parent > 0 ? (parent - 1) >> 1 : 0;
            uint4 p, g;
            if (PushPrefetch<K>::ON && parent == pre.idx) {              // (pre.idx >= heap_lds > 0, its parent in LDS: timer_push_prefetch)
#ifdef MADSIM_EMU
                { const uint2 now = nh_load<K>(c, parent); if (now.x != pre.e.x || now.y != pre.e.y) OVF_SET(L, OVF_BUG); }    // nothing touched the heap since
#endif
                p = nh_expand(L, pre.e);
                g = nh_expand(L, LDS64(c.heap0 + (gp << LWSH<K>(c))));
                pre.idx = ~0u;
            }
            else if (parent > 0) heap_get2<K>(c, L, gp, parent, g, p);
            else { p = heap_get<K>(c, L, 0); g = p; }
            up = hd < (parent == 0 ? L.top_dl : ev_deadline(p));
            if (up) {
                heap_set<K>(c, L, pos, p); pos = parent;
                up = pos > 0 && hd < (gp == 0 ? L.top_dl : ev_deadline(g));
                if (up) { heap_set<K>(c, L, pos, g); pos = gp; up = pos > 0; }
            }
        } else {
            const uint4 p = heap_get<K>(c, L, parent);
            // hole <= parent in heap order: stop (the root's deadline is mirrored in a register)
            up = hd < (parent == 0 ? L.top_dl : ev_deadline(p));
            if (up) { heap_set<K>(c, L, pos, p); pos = parent; up = pos > 0; }
        }
    }
    heap_set<K>(c, L, pos, hole);
    if (pos == 0) L.top_dl = hd;
}

// Timer::add -> BinaryHeap::push.  Returns false on capacity overflow.
template <class K>
__device__ __forceinline__ bool timer_add(const Ctx& c, Lane& L, uint64_t deadline, uint32_t meta, uint32_t val, HeapPre pre = HeapPre{~0u, make_uint2(0, 0)}) {
    PROBE2(0);
    REG(10);
    bool room = L.heap_len < c.P.heap_lds + (K::SPILL ? c.P.heap_spill : 0u);
    if (K::NH) {
        // the 8-byte entry keeps the low deadline word: a deadline 2^31 ns or more ahead of the clock cannot be told from an earlier one.
        // A capacity verdict like a full heap: the re-run leaves the narrow layout (madsim_hip.cpp grow)
        if (deadline - L.clock >= NH_HORIZON) room = false;
        if (room && (meta >> EV_SHIFT) == EV_DELIVER) {      // tag, sender and payload wait in a pool record until the entry is the root
            const uint32_t r = pool_alloc<K>(c);
            if (r == ~0u) room = false;
            else {
                buf_store64(c.gs, pool_addr(c, r), make_uint2(meta, val));
                meta = (meta & 0xffe0003fu) | (r << 6);
            }
        }
    }
    if (room) {                                        // (no early return: see k_main.h on exit edges)
        uint4 e = make_uint4((uint32_t)deadline, (uint32_t)(deadline >> 32), meta, val);
        heap_sift_up<K>(c, L, L.heap_len, e, pre);
        L.heap_len++;
    }
    PROBE2(10);
    return room;
}

// ---- MADSIM_STATE_DEDUP_TIMERS (Variant::DEDUP builds, KParams.dedup_n != 0) ------------------------------------------------
// Sleep::poll registers ANOTHER timer with the same deadline and the same waker on every not-elapsed poll (time/sleep.rs:51-53);
// the first one is still in the heap then (it fires once deadline <= now, and the poll saw now < deadline), and when the
// deadline comes all of them fire back to back, every one after the first finding its task SCHEDULED already: one executor step
// each, nothing else.  In the election loop a third of all Timer::add calls are such repeats and they make up half of the heap,
// most of it in the HBM spill region.  Here a repeat becomes a count in a small per-seed hash table — dedup_n 16-byte buckets
// {deadline lo, hi, wake meta, count} behind the task units — and the pop of the first entry with that (deadline, meta) adds
// the count to the step counter.  A bucket held by another key: the repeat is pushed like everywhere else (always right).
// What the shorter heap cannot reproduce is the order of two DIFFERENT entries with equal deadlines (the array algorithm's tie
// order depends on the heap's shape): timer_expire notices every such tie as it pops it (Lane::hazard) and the seed starts over
// with Lane::exact set (k_main.h) — results never differ.
__device__ __forceinline__ uint32_t dedup_index(const Ctx& c, uint64_t deadline, uint32_t meta) {
    const uint32_t lo = (uint32_t)deadline;
    const uint32_t h = lo ^ (lo >> 7) ^ (lo >> 15) ^ ((uint32_t)(deadline >> 32) * 0x9e3779b1u) ^ ((meta & 0xffu) * 0x85ebca6bu);
    return (h ^ (h >> 11)) & (c.P.dedup_n - 1u);
}
__device__ __forceinline__ uint32_t dedup_at(const Ctx& c, uint32_t idx) { return c.P.dedup_off + idx * 16u; }   // logical byte offset of the bucket's unit
// Which buckets hold a count is mirrored in a register (Lane::dd_occ, bit = bucket index; dedup_n <= 64): an EMPTY bucket is neither
// loaded by the re-registration that fills it nor by the pop of a wake-up that maps to it (MADSIM_DEDUP_OCC; round 6: half of both kinds).
// A re-registration of (deadline, meta): true = it now lives in the table; false = push it.
__device__ __forceinline__ bool dedup_note(const Ctx& c, Lane& L, uint64_t deadline, uint32_t meta) {
    const uint32_t idx = dedup_index(c, deadline, meta), at = dedup_at(c, idx);
    bool noted = false;
    if (MADSIM_DEDUP_OCC && !((L.dd_occ >> idx) & 1ull)) {
#ifdef MADSIM_EMU
        if (gs_load128(c.gs, gs_addr_unit(c, at)).w != 0) OVF_SET(L, OVF_BUG);        // the mirror says empty: it is
#endif
        gs_store128(c.gs, gs_addr_unit(c, at), make_uint4((uint32_t)deadline, (uint32_t)(deadline >> 32), meta, 1u));
        L.dd_occ |= 1ull << idx;
        noted = true;
    } else {
        const uint4 u = gs_load128(c.gs, gs_addr_unit(c, at));
        const bool same = u.x == (uint32_t)deadline && u.y == (uint32_t)(deadline >> 32) && u.z == meta;
#ifdef MADSIM_EMU
        if (MADSIM_DEDUP_OCC && u.w == 0) OVF_SET(L, OVF_BUG);                          // the mirror says occupied: it is
#endif
        if (u.w == 0) { gs_store128(c.gs, gs_addr_unit(c, at), make_uint4((uint32_t)deadline, (uint32_t)(deadline >> 32), meta, 1u)); noted = true; }
        else if (same && u.w != ~0u) { gs_store32(c.gs, gs_addr_uword(c, at + 12u), u.w + 1u); noted = true; }
    }
    return noted;
}

// Timer::add as the executor code calls it.  Every build but the global-state ones pushes at once; those queue the call in
// the lane (k_state.h Lane::pq_*) until timer_flush.  `wake` = the event is a wake-up of the task being polled (meta is the
// same for all of them, so only the deadline is kept); a delivery always precedes the wake-ups of its round (k_poll.h).
template <class K>
// `again` = the call re-registers a Sleep whose first timer is still in the heap (DEDUP builds: see dedup_note).
__device__ __forceinline__ void timer_schedule(const Ctx& c, Lane& L, uint64_t deadline, uint32_t meta, uint32_t val, bool wake, bool again = false) {
    if (!K::G) { if (!timer_add<K>(c, L, deadline, meta, val)) OVF_SET(L, OVF_CAP); return; }
    if (wake) {
        const uint32_t n = L.pq_n & 7u;
        if (n >= 3) OVF_SET(L, OVF_BUG);                            // (cannot happen: a round makes at most three; never a lost timer)
        else {
            // (value selects, not `if (n == 0) L.pq_w0 = ..`: a store through a selected field address keeps the whole Lane in scratch)
            L.pq_w0 = n == 0 ? deadline : L.pq_w0; L.pq_w1 = n == 1 ? deadline : L.pq_w1; L.pq_w2 = n == 2 ? deadline : L.pq_w2;
            L.pq_n++;
            if (K::DEDUP && again) L.pq_n |= 8u << n;
        }
    } else {
        if (L.pq_n) OVF_SET(L, OVF_BUG);                            // (cannot happen: one delivery per round, before its wake-ups)
        L.pq_deliv_dl = deadline; L.pq_deliv_meta = meta; L.pq_deliv_val = val; L.pq_n |= 0x80u;
    }
}
// Perform the queued pushes, oldest first.  `wake_meta` = the wake-up event of the task being polled.
// The request a poll round makes for its first push (k_poll.h): the parent of slot heap_len, when that is a spilled entry whose own parent is in LDS
// (the topology's heap: 74 entries over 31 LDS slots — a push starts on level 6, its parent on level 5, the grandparent on level 4).
template <class K>
__device__ __forceinline__ HeapPre timer_push_prefetch(const Ctx& c, const Lane& L) {
    HeapPre pre = {~0u, make_uint2(0, 0)};
    if (PushPrefetch<K>::ON) {
        const uint32_t cap = c.P.heap_lds, pos = L.heap_len;
        if (pos >= 2 * cap + 1 && pos < 4 * cap + 3) {          // parent = (pos - 1) / 2 >= cap, grandparent = (parent - 1) / 2 < cap
            const uint32_t parent = (pos - 1) >> 1;
            pre.e = buf_load64(c.spill, (parent - cap) * c.P.total_lanes * 8u + c.spill_off);
            pre.idx = parent;
        }
    }
    return pre;
}
template <class K>
__device__ __forceinline__ void timer_flush(const Ctx& c, Lane& L, uint32_t wake_meta, HeapPre pre = HeapPre{~0u, make_uint2(0, 0)}) {
    if (!K::G) return;
    while (L.pq_n) {
        uint64_t dl; uint32_t meta, val;
        if (L.pq_n & 0x80u) { dl = L.pq_deliv_dl; meta = L.pq_deliv_meta; val = L.pq_deliv_val; L.pq_n &= 0x7fu; }
        else if (!K::DEDUP) { dl = L.pq_w0; meta = wake_meta; val = 0; L.pq_w0 = L.pq_w1; L.pq_w1 = L.pq_w2; L.pq_n--; }
        else {
            dl = L.pq_w0; meta = wake_meta; val = 0; L.pq_w0 = L.pq_w1; L.pq_w1 = L.pq_w2;
            const bool again = (L.pq_n & 8u) != 0;
            L.pq_n = ((L.pq_n & 7u) - 1u) | ((L.pq_n >> 1) & 0x18u);        // one wake-up less; the repeat flags move down with their deadlines
            if (again && c.P.dedup_n && !L.exact && dedup_note(c, L, dl, meta)) continue;
        }
        if (!timer_add<K>(c, L, dl, meta, val, pre)) OVF_SET(L, OVF_CAP);
        pre.idx = ~0u;                                      // (the heap has changed: the request was for the round's first push)
    }
}

// BinaryHeap::pop: swap the last element into the root, sift_down_to_bottom(0), then sift_up.
template <class K>
__device__ __forceinline__ uint4 timer_pop(const Ctx& c, Lane& L) {
#ifdef MADSIM_EMU
    if (K::G && L.pq_n) OVF_SET(L, OVF_BUG);      // a pop with Timer::add calls of the round still queued (see timer_expire, k_net.h): MADSIM_INTERNAL in the emulation suite
#endif
    PROBE2(0);
    REG(20);
    uint32_t end = --L.heap_len;
    uint4 item = heap_get<K>(c, L, end);
    if (end > 0) {
        uint4 top = heap_get<K>(c, L, 0);
        uint32_t pos = 0, child = 1;
        uint4 m = item;                                     // the entry last moved up: it now sits at parent(pos)
        if (MADSIM_POP_TOPDOWN && K::SPILL) {
            // The same final array top-down.  sift_down_to_bottom moves the smaller child up on every level of the path (the right one on a tie), then
            // sift_up carries `item` back up past every path entry with a deadline GREATER than its own — the path's deadlines do not decrease
            // downwards, so those are a suffix of it and every one of them returns to the slot it came from.  Stopping at the first path entry that
            // is greater and storing `item` there writes exactly the slots that end up different, and skips the loads of the levels below and
            // the stores that would be undone.  (Measured: election loop +2 %, timer storm +2.2 %, topology +0.1 %; the KV's build — no spill region, a heap of a few LDS
            // entries — read 0.3 % slower with it and keeps the literal form.  Also built: two spilled levels per global round trip, the children's
            // children loaded with them — topology and election loop both -0.8 %: the extra lane accesses cost more than the round trip.
            // profiles/r6_experiments.md)
            const uint64_t idl = ev_deadline(item);
            // MADSIM_POP_LDS_FIRST: `item` is the heap's LAST entry — with any depth a load from the spill region — and the stop test of every level
            // waits for it: the walk through the LDS-resident levels stood behind a global round trip it does not need.  Those levels are
            // walked the literal way instead (sift_down_to_bottom: the smaller child moves up, no look at `item`), which brings the walk to the
            // first spilled level while the load is still in flight: that level's children are requested beside it, one round trip instead of
            // two.  The spilled levels keep the stop test.  If nothing moved there, `item` may belong above the LDS levels' last slot: the literal
            // sift_up from it (parents in LDS; the common case — the old bottom entry belongs at the bottom again — compares with the entry in
            // hand and loads nothing).  Same comparisons on the same values as sift_down_to_bottom + sift_up: the same array.
            bool moved_below = false;
            if (MADSIM_POP_LDS_FIRST && !K::CMP) {
                const uint32_t cap = c.P.heap_lds;
                while (child + 1 < end && child + 1 < cap) {
                    REG(21);
                    uint4 l, r;
                    heap_lds_get2<K>(c, L, child, l, r);
                    const bool right = ev_deadline(l) >= ev_deadline(r);
                    m = right ? r : l;
                    heap_set_lds<K>(c, L, pos, m);
                    if (pos == 0) L.top_dl = ev_deadline(m);
                    pos = child + (right ? 1u : 0u);
                    child = 2 * pos + 1;
                }
            }
            bool go = child + 1 < end, stopped = false;
            while (go) {                                   // (one exit: see k_main.h)
                REG(21);
                uint4 l, r;
                heap_get2<K>(c, L, child, child + 1, l, r);
                const bool right = ev_deadline(l) >= ev_deadline(r);
                const uint4 cand = right ? r : l;
                go = ev_deadline(cand) <= idl;
                if (go) {
                    m = cand;
                    heap_set<K>(c, L, pos, m);
                    if (pos == 0) L.top_dl = ev_deadline(m);
                    pos = child + (right ? 1u : 0u);
                    child = 2 * pos + 1;
                    go = child + 1 < end;
                    moved_below = true;
                } else stopped = true;
            }
            if (!stopped && child == end - 1) {
                const uint4 cand = heap_get<K>(c, L, child);
                if (ev_deadline(cand) <= idl) {
                    m = cand;
                    heap_set<K>(c, L, pos, m);
                    if (pos == 0) L.top_dl = ev_deadline(m);
                    pos = child;
                    moved_below = true;
                }
            }
            // (`m` = the entry last moved up: it sits at parent(pos).  Moved under the stop test it is <= item, and so is everything above it.)
            if (MADSIM_POP_LDS_FIRST && !K::CMP && !moved_below && pos > 0 && idl < ev_deadline(m)) {
                heap_set<K>(c, L, pos, m); pos = (pos - 1) >> 1; heap_sift_up<K>(c, L, pos, item);
            } else {
                heap_set<K>(c, L, pos, item);
                if (pos == 0) L.top_dl = idl;
            }
        } else {
        while (child + 1 < end) {
            REG(21);
            uint4 l, r;
            heap_get2<K>(c, L, child, child + 1, l, r);
            bool right = ev_deadline(l) >= ev_deadline(r);   // left <= right in heap order: take right
            m = right ? r : l;
            heap_set<K>(c, L, pos, m);
            if (pos == 0) L.top_dl = ev_deadline(m);
            pos = child + (right ? 1u : 0u);
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            m = heap_get<K>(c, L, child);
            heap_set<K>(c, L, pos, m);
            if (pos == 0) L.top_dl = ev_deadline(m);
            pos = child;
        }
        // sift_up(0, pos) of `item`.  Its first comparison is with the entry at parent(pos) — the one just moved there, still
        // in registers — so the common outcome (the old bottom element belongs at the bottom again) costs no load: on the
        // spill levels that is one dependent global round trip less per pop.  (Same comparisons, same stores as
        // heap_sift_up from `pos`: only where the parent's value comes from differs.)
        if (K::SPILL && pos > 0) {
            if (ev_deadline(item) < ev_deadline(m)) { heap_set<K>(c, L, pos, m); pos = (pos - 1) >> 1; heap_sift_up<K>(c, L, pos, item); }
            else heap_set<K>(c, L, pos, item);
        } else {
            heap_sift_up<K>(c, L, pos, item);
        }
        }
        item = top;
    } else {
        L.top_dl = ~0ull;
    }
    PROBE2(11);
    return item;
}

}  // namespace madsim_k

#endif
