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
__device__ __forceinline__ uint4 spill_load(const Ctx& c, uint32_t slot) {
    return buf_load128(c.spill, slot * c.P.total_lanes * 16u + c.spill_off);
}
__device__ __forceinline__ void spill_store(const Ctx& c, uint32_t slot, const uint4& e) {
    buf_store128(c.spill, slot * c.P.total_lanes * 16u + c.spill_off, e);
}

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
template <class K>
__device__ __forceinline__ uint4 heap_get(const Ctx& c, uint32_t i) {
    if (!K::SPILL) return heap_lds_get<K>(c, i);
    uint32_t cap = c.P.heap_lds;
    uint4 v = heap_lds_get<K>(c, i < cap ? i : cap - 1);
    if (i >= cap) v = spill_load(c, i - cap);
    return v;
}
template <class K>
__device__ __forceinline__ void heap_set(const Ctx& c, uint32_t i, const uint4& e) {
    if (!K::SPILL || i < c.P.heap_lds) heap_lds_set<K>(c, i, e);
    else spill_store(c, i - c.P.heap_lds, e);
}
// meta + payload of a datagram delivery event (net/mod.rs:323-330): destination socket `ds` of incarnation `sgen`.
template <class K>
__device__ __forceinline__ uint2 ev_deliver_meta(uint32_t sgen, uint32_t tag, uint32_t from, uint32_t ds, uint32_t val, uint32_t pc) {
    if (K::LIFE) return make_uint2((EV_DELIVER << EV_SHIFT) | (sgen << 21) | (tag << 13) | (from << 6) | ds, val);
    return make_uint2((EV_DELIVER << EV_SHIFT) | (sgen << 21) | (pc << 6) | ds, 0);     // base ops: tag, from, payload = fields of insn pc
}

// BinaryHeap::sift_up(0, pos) with `hole` as the moving element; keeps the root mirror current.
template <class K>
__device__ __forceinline__ void heap_sift_up(const Ctx& c, Lane& L, uint32_t pos, const uint4& hole) {
    uint64_t hd = ev_deadline(hole);
    bool up = pos > 0;
    while (up) {                                       // (one exit: see k_main.h)
        REG(11);
        const uint32_t parent = (pos - 1) >> 1;
        const uint4 p = heap_get<K>(c, parent);
        // hole <= parent in heap order: stop (the root's deadline is mirrored in a register)
        up = hd < (parent == 0 ? L.top_dl : ev_deadline(p));
        if (up) { heap_set<K>(c, pos, p); pos = parent; up = pos > 0; }
    }
    heap_set<K>(c, pos, hole);
    if (pos == 0) L.top_dl = hd;
}

// Timer::add -> BinaryHeap::push.  Returns false on capacity overflow.
template <class K>
__device__ __forceinline__ bool timer_add(const Ctx& c, Lane& L, uint64_t deadline, uint32_t meta, uint32_t val) {
    PROBE2(0);
    REG(10);
    const bool room = L.heap_len < c.P.heap_lds + (K::SPILL ? c.P.heap_spill : 0u);
    if (room) {                                        // (no early return: see k_main.h on exit edges)
        uint4 e = make_uint4((uint32_t)deadline, (uint32_t)(deadline >> 32), meta, val);
        heap_sift_up<K>(c, L, L.heap_len, e);
        L.heap_len++;
    }
    PROBE2(10);
    return room;
}

// BinaryHeap::pop: swap the last element into the root, sift_down_to_bottom(0), then sift_up.
template <class K>
__device__ __forceinline__ uint4 timer_pop(const Ctx& c, Lane& L) {
    PROBE2(0);
    REG(20);
    uint32_t end = --L.heap_len;
    uint4 item = heap_get<K>(c, end);
    if (end > 0) {
        uint4 top = heap_get<K>(c, 0);
        uint32_t pos = 0, child = 1;
        while (child + 1 < end) {
            REG(21);
            uint4 l = heap_get<K>(c, child), r = heap_get<K>(c, child + 1);
            bool right = ev_deadline(l) >= ev_deadline(r);   // left <= right in heap order: take right
            uint4 m = right ? r : l;
            heap_set<K>(c, pos, m);
            if (pos == 0) L.top_dl = ev_deadline(m);
            pos = child + (right ? 1u : 0u);
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            uint4 m = heap_get<K>(c, child);
            heap_set<K>(c, pos, m);
            if (pos == 0) L.top_dl = ev_deadline(m);
            pos = child;
        }
        heap_sift_up<K>(c, L, pos, item);
        item = top;
    } else {
        L.top_dl = ~0ull;
    }
    PROBE2(11);
    return item;
}

}  // namespace madsim_k

#endif
