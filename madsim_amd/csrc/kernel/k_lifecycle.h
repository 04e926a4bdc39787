// k_lifecycle.h — Task spawn / finish and node lifecycle (kill, restart, pause).
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_LIFECYCLE_H
#define MADSIM_K_LIFECYCLE_H

namespace madsim_k {

// ---- task lifecycle ----------------------------------------------------------------------------
// `via_handle`: spawned through a NodeHandle captured at build() (NodeHandle::spawn from another node's task:
// that Spawner holds the ORIGINAL Arc<NodeInfo>); otherwise task::spawn / init on the node's current info.
// `init` (MS_OP_SPAWN from the polled task, k_poll.h): words of the spawning task its poll holds in registers — its flag word and its
// unit 1 word 1 (spawn order | info_gen << 24: the NodeInfo it runs under), so task::spawn reads neither — and what `async move` hands the
// child: the request's value and sender (unit 0 w and y bits 24-31) and the request word of the rpc unit, stored WITH the new task's units
// instead of over them afterwards.
// `words_known` (every-class global-state builds, k_poll.h stage [C]): the free slot and the words the spawn reads — the slot's old flag word, the
// node's info-generation word, the spawn counter, the gen-0 killed mask of a NodeHandle::spawn — were requested with the stage's other reads.
struct SpawnInit { bool parent_known; uint32_t parent_f, parent_u1y; bool move_req; uint32_t w, from, req;
                   bool words_known; uint32_t slot, w_old, w_gen, seq, w_killed; };
template <class K>
__device__ __forceinline__ uint32_t spawn_task(const Ctx& c, Lane& L, uint32_t prog, bool record, bool via_handle = false, int parent = -1,
                                               const SpawnInit init = SpawnInit{false, 0, 0, false, 0, 0, 0, false, 0, 0, 0, 0, 0}) {
    uint32_t slot = 0;
    if (K::G) {                                              // first free slot = first zero bit of the alive mask (LDS)
        slot = c.P.max_tasks;
        if (init.words_known) slot = init.slot;              // (found by the caller, nothing spawned since)
        else for (uint32_t wi = 0; wi < (c.P.max_tasks + 31) / 32; wi++) {
            uint32_t free_bits = ~AMASK(wi);
            if (free_bits) { slot = wi * 32 + (uint32_t)__builtin_ctz(free_bits); break; }
        }
#ifdef MADSIM_EMU
        if (init.words_known && ((AMASK(slot >> 5) >> (slot & 31)) & 1)) OVF_SET(L, OVF_BUG);
#endif
    } else {
        while (slot < c.P.max_tasks && (TWORD(c, slot, 0, 0) & TF_ALIVE)) slot++;
    }
    // (the task table holds max_tasks <= 254 live tasks — an 8-bit slot.  Short of that a larger limit lifts it: a capacity verdict, the seed is
    //  run again.  AT 254 nothing does: the 255th live task leaves the workload model, MADSIM_UNSUPPORTED — and since the oracle fills its
    //  unbounded Vec lowest free slot first, as this does, it says so at the same spawn)
    if (slot >= c.P.max_tasks) { OVF_SET(L, c.P.max_tasks >= MADSIM_MAX_LIVE_TASKS ? OVF_MODEL : OVF_CAP); return 0xffffffffu; }
    if (K::G) AMASK(slot >> 5) |= 1u << (slot & 31);
    uint32_t pw, node, gen, killed = 0, info_gen = 0, seq = 0;
    if constexpr (K::G) {
        pw = PROGW(c, prog);
        node = pw & 0xff;
        // Every word the spawn reads is requested here, before the first is looked at (global-state builds: the old slot's generation, the
        // node's info generation, the killed mask or the parent's words, the spawn counter came one dependent round trip after another —
        // three for a task::spawn — with the whole wave waiting on each).  What the branches below do not use is not used; nothing is
        // written in between.
        uint32_t w_old = init.w_old;
        // (raw words only up to the last request: arithmetic on a loaded value inside a divergent block makes the compiler wait for it there)
        uint32_t w_gen = init.w_gen, w_parent_f = init.parent_f, w_parent_y = init.parent_u1y, w_killed = init.w_killed;
        const bool in_parent = parent >= 0;
        seq = init.seq;
        if (!init.words_known) {
            w_old = TWORD(c, slot, 0, 0);
            if (K::FN) {
                w_gen = NODEW(4 + (node >> 2));                  // NODE_INFO_GEN(node)
                seq = NODEW(3);
                if (!in_parent) w_killed = NODEW(via_handle ? 2u : 0u);
            }
        }
#ifdef MADSIM_EMU
        if (init.words_known && (w_old != (uint32_t)TWORD(c, slot, 0, 0) || (K::FN && (w_gen != (uint32_t)NODEW(4 + (node >> 2)) || seq != (uint32_t)NODEW(3) ||
                                  (!in_parent && w_killed != (uint32_t)NODEW(via_handle ? 2u : 0u)))))) OVF_SET(L, OVF_BUG);
#endif
        if (K::FN) {
            if (in_parent && !init.parent_known) { w_parent_f = TWORD(c, (uint32_t)parent, 0, 0); w_parent_y = TWORD(c, (uint32_t)parent, 1, 1); }
#ifdef MADSIM_EMU
            if (in_parent && init.parent_known && (((uint32_t)TWORD(c, (uint32_t)parent, 0, 0) ^ init.parent_f) & TF_KILLED)) OVF_SET(L, OVF_BUG);     // the poll's copy is current
            if (in_parent && init.parent_known && (uint32_t)TWORD(c, (uint32_t)parent, 1, 1) != init.parent_u1y) OVF_SET(L, OVF_BUG);
#endif
        }
        gen = (((w_old >> 8) & 0xffff) + 1) & 0xffff;
        if (K::FN) {
            const uint32_t cur_gen = (w_gen >> ((node & 3) * 8)) & 0xff;
            info_gen = cur_gen;
            if (in_parent) {             // task::spawn inside that task's context: its own Arc<NodeInfo>, whatever became of the node
                killed = (w_parent_f & TF_KILLED) ? 1u : 0u;
                info_gen = w_parent_y >> 24;
            }
            else if (via_handle && cur_gen != 0) { killed = 1; info_gen = 0; }        // stale handle: dead info
            else killed = (w_killed >> node) & 1;                                      // task/mod.rs:632-634
            NODEW(3) = seq + 1;                                  // spawn order matters only to NodeInfo::kill
        }
    } else {
        // (LDS-resident builds: the reads are LDS reads, in the order the code has always had)
        gen = (((TWORD(c, slot, 0, 0) >> 8) & 0xffff) + 1) & 0xffff;
        pw = PROGW(c, prog);
        node = pw & 0xff;
        if (K::FN) {
            uint32_t cur_gen = NODE_INFO_GEN(node);
            info_gen = cur_gen;
            if (parent >= 0) {           // task::spawn inside that task's context: its own Arc<NodeInfo>, whatever became of the node
                killed = (TWORD(c, (uint32_t)parent, 0, 0) & TF_KILLED) ? 1u : 0u;
                info_gen = TWORD(c, (uint32_t)parent, 1, 1) >> 24;
            }
            else if (via_handle && cur_gen != 0) { killed = 1; info_gen = 0; }        // stale handle: dead info
            else killed = ((via_handle ? NODEW(2) : NODEW(0)) >> node) & 1;          // task/mod.rs:632-634
        }
        if (K::FN) { seq = NODEW(3); NODEW(3) = seq + 1; }      // spawn order matters only to NodeInfo::kill
    }
    // (MS_OP_SPAWN with move_request: `async move` takes the request — rpc.rs:170 — its value, its sender and the rsp_tag word)
    TU(c, slot, 0) = make_uint4(TF_ALIVE | TF_SCHED | (killed ? TF_KILLED : 0) | (gen << 8) | (prog << 24),
                                init.move_req ? ((pw >> 16) & 0x00ffffffu) | (init.from << 24) : pw >> 16, 0, init.move_req ? init.w : 0u);
    tu1_store<K>(c, slot, make_uint4(0xffu << 8, (seq & 0xffffff) | (info_gen << 24), 0, 0));   // rxseq 0, no awaiter; spawn order
    if (K::FC && c.P.uses_chan) TU(c, slot, c.P.chan_unit) = make_uint4(0xff, 0, 0, 0);                // no connection held
    if (K::FR && c.P.uses_rpc) { TWORD(c, slot, c.P.rpc_unit, 0) = init.move_req ? init.req : 0u; TWORD(c, slot, c.P.rpc_unit, 1) = 0; }   // no request in hand
    ready_push<K>(c, L, slot);
    if (record) HW(prog) = H_RUNNING | (slot << 8) | (gen << 16);
    return slot;
}

template <class K> __device__ __forceinline__ void conn_drop_handles(const Ctx& c, Lane& L, uint32_t id, uint32_t side, bool node_killed);
template <class K> __device__ __forceinline__ void endpoint_drop(const Ctx& c, Lane& L, uint32_t s, bool node_killed);
template <class K> __device__ __forceinline__ void sock_drop_acceptq(const Ctx& c, Lane& L, uint32_t s);

// The locals of a task body drop: its (Sender, Receiver) pair, then the Endpoints it holds (table order).  `f` = its flag word.
// (`known` / `cx_known`: the caller holds word 0 of the task's connection unit — the task that finishes inside its own poll)
template <class K>
__device__ __forceinline__ void task_drop_locals(const Ctx& c, Lane& L, uint32_t slot, uint32_t f, bool known = false, uint32_t cx_known = 0) {
    const uint32_t gen = (f >> 8) & 0xffff;
    if (K::FC && c.P.uses_chan) {
        uint32_t cx = known ? cx_known : (uint32_t)TWORD(c, slot, c.P.chan_unit, 0);
        if ((cx & 0xff) != 0xff) { conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1, (f & TF_KILLED) != 0); TWORD(c, slot, c.P.chan_unit, 0) = cx | 0xff; }
    }
    if (K::G && (f & TF_OWNER)) {
        // Global-state builds: the candidates are the sockets bound by a task that is still around (the owner mask, LDS); their owner
        // words are read four to a round trip — a finishing task used to walk them one dependent read at a time, the wave waiting
        // on each.  Table order is kept; dropping one Endpoint writes no other socket's owner word.
        const uint32_t key = slot | (gen << 16);
        for (uint32_t wi = 0; wi < 2; wi++) {
            uint32_t m = OMASK(wi);
            while (m) {
                uint32_t ix[4], ow[4], n = 0;
                for (uint32_t k = 0; k < 4; k++) { ix[k] = wi * 32 + (uint32_t)__builtin_ctz(m | 0x80000000u); if (m) { n++; m &= m - 1; } }
                for (uint32_t k = 0; k < 4; k++) ow[k] = SW(c, k < n ? ix[k] : ix[0], 1);
                for (uint32_t k = 0; k < 4; k++) {
                    if (k >= n || ow[k] != key) continue;
                    const uint32_t i = ix[k];
                    OMASK(i >> 5) &= ~(1u << (i & 31));
                    endpoint_drop<K>(c, L, i, (f & TF_KILLED) != 0);
                    if (SW(c, i, 1) != ~0u) SW(c, i, 1) = 0x0000ff00u;    // nobody's: no owner word has bits 8-15 set
                }
            }
        }
    } else if (f & TF_OWNER) {                                // LDS-resident builds: every table entry, in order
        for (uint32_t i = 0; i < c.P.n_socks; i++) {
            const uint32_t hdr = SW(c, i, 0);
            if (!sock_owned_by<K>(c, i, hdr, slot, gen)) continue;
            endpoint_drop<K>(c, L, i, (f & TF_KILLED) != 0);
            if (K::LIFE && SW(c, i, 1) != ~0u) SW(c, i, 1) = 0x0000ff00u;    // nobody's: no owner word has bits 8-15 set
        }
    }
}

// MADSIM_PROG_DROP_SPAWN: the guard moved into the body drops after its other locals, and its Drop spawns program prog + 1 in
// this task's context (task/mod.rs:1185-1253).  Full builds only (geometry.h routes such workloads there, like address resolution).
template <class K>
__device__ __forceinline__ void task_drop_guard(const Ctx& c, Lane& L, uint32_t slot, uint32_t prog) {
    if (K::FN && K::FA && ((PROGW(c, prog) >> 8) & MADSIM_PROG_DROP_SPAWN)) spawn_task<K>(c, L, prog + 1, false, false, (int)slot);
}

// The future is gone: completed (outcome H_COMPLETED) or dropped by the executor (H_CANCELLED).
// `kn` (global-state builds, the task that completes inside its own poll — MS_OP_DONE): the words of the task the poll holds in
// registers and has just written back — flag word, awaiter link (unit 1 word 0), connection word — instead of three dependent reads.
// (`have_h`: the JoinHandle word too — requested with the stage's other reads, k_poll.h stage [C])
struct FinishKnown { bool have; uint32_t f, link, cx; bool have_h; uint32_t h; };
template <class K>
__device__ __forceinline__ uint32_t task_finish(const Ctx& c, Lane& L, uint32_t slot, uint32_t outcome, bool guard = true, FinishKnown kn = FinishKnown{false, 0, 0, 0, false, 0}) {
    uint32_t f = kn.have ? kn.f : (uint32_t)TWORD(c, slot, 0, 0);
    uint32_t gen = (f >> 8) & 0xffff, prog = f >> 24;
    const uint32_t h_pre = kn.have ? (kn.have_h ? kn.h : (uint32_t)HW(prog)) : 0u;       // (requested before the locals drop: it arrives with their first read)
#ifdef MADSIM_EMU
    if (kn.have && kn.have_h && kn.h != (uint32_t)HW(prog)) OVF_SET(L, OVF_BUG);
#endif
    task_drop_locals<K>(c, L, slot, f, kn.have, kn.cx);
    if (guard) task_drop_guard<K>(c, L, slot, prog);
    // (h_pre stays valid: between its read and here only this task's own locals dropped — no JoinHandle word is written by that)
    uint32_t h = kn.have && !(K::FN && K::FA) ? h_pre : (uint32_t)HW(prog);
    if (h == (H_RUNNING | (slot << 8) | (gen << 16))) { HW(prog) = (h & ~3u) | outcome; if (prog == 0) L.main_done = 1; }
    uint32_t link = kn.have ? kn.link : (uint32_t)TWORD(c, slot, 1, 0);
    TWORD(c, slot, 0, 0) = f & ~(TF_ALIVE | TF_SCHED | TF_RUN | TF_INBOX);
    if (K::G) AMASK(slot >> 5) &= ~(1u << (slot & 31));
    uint32_t j = (link >> 8) & 0xff;
    if (j != 0xff) {                                          // async-task hands the output over and notifies the awaiter
        const uint32_t jf = TWORD(c, j, 0, 0);
        if ((jf & TF_ALIVE) && ((jf >> 8) & 0xffff) == (link >> 16)) {     // still that task: it is parked in its MS_OP_JOIN
            const uint32_t jy = TWORD(c, j, 0, 1);
            if (((jy >> 16) & 0xff) == SUB_JOIN_WAIT) TWORD(c, j, 0, 1) = (jy & ~0x00ff0000u) | ((outcome == H_CANCELLED ? SUB_JOIN_CANCELLED : SUB_JOIN_COMPLETED) << 16);
        }
        wake<K>(c, L, j, link >> 16);
    }
    return f & ~(TF_ALIVE | TF_SCHED | TF_RUN | TF_INBOX);       // the flag word as this call left it
}

// NodeInfo::kill (task/mod.rs:133-140): mark + wake every live task holding NodeInfo `info_gen` of `node`, in
// spawn order (the order of NodeInfo.tasks).  Tasks carry their spawn sequence number, so no list is stored.
template <class K>
__device__ __forceinline__ void info_kill(const Ctx& c, Lane& L, uint32_t node, uint32_t info_gen) {
    if constexpr (K::G) {
        // Global-state builds.  The walk below reads a task's flag word, looks at it, reads its unit 1, looks at it — two dependent round trips
        // per live task, the whole list once per task it kills: ~60 round trips for the topology's kill + restart, and with 64 seeds reaching
        // theirs in different passes the wave sat in this loop for 2.6 round trips of EVERY pass (tools/mem_site_model.py topo).  Here: the live
        // tasks (alive mask, LDS) four at a time, both words of the four requested together; the matches kept as the four smallest
        // (spawn order << 8 | slot) of one scan, killed in that order with their flag words read together; another scan only when more than four
        // matched.  The same tasks get TF_KILLED and their wake-up in the same order.
        uint32_t last = 0; bool any_last = false;
        for (;;) {
            uint32_t s0 = ~0u, s1 = ~0u, s2 = ~0u, s3 = ~0u; bool more = false;
            for (uint32_t wi = 0; wi < (c.P.max_tasks + 31) / 32; wi++) {
                uint32_t m = AMASK(wi);
                while (m) {
                    uint32_t ix[4], fw[4], sw[4], n = 0;
                    for (uint32_t k = 0; k < 4; k++) { ix[k] = wi * 32 + (uint32_t)__builtin_ctz(m | 0x80000000u); if (m) { n++; m &= m - 1; } }
                    for (uint32_t k = 0; k < 4; k++) { fw[k] = TWORD(c, k < n ? ix[k] : ix[0], 0, 0); sw[k] = TWORD(c, k < n ? ix[k] : ix[0], 1, 1); }
                    for (uint32_t k = 0; k < 4; k++) {
                        if (k >= n || !(fw[k] & TF_ALIVE) || (PROGW(c, fw[k] >> 24) & 0xff) != node || (sw[k] >> 24) != info_gen) continue;
                        uint32_t t = (sw[k] << 8) | ix[k], u;           // spawn order (24 bits, unique) | slot
                        if (any_last && t <= last) continue;
                        u = t < s0 ? t : s0; t = t < s0 ? s0 : t; s0 = u;
                        u = t < s1 ? t : s1; t = t < s1 ? s1 : t; s1 = u;
                        u = t < s2 ? t : s2; t = t < s2 ? s2 : t; s2 = u;
                        u = t < s3 ? t : s3; t = t < s3 ? s3 : t; s3 = u;
                        if (t != ~0u) more = true;                      // a fifth match: it waits for the next scan
                    }
                }
            }
            const uint32_t f0 = s0 != ~0u ? (uint32_t)TWORD(c, s0 & 0xff, 0, 0) : 0u, f1 = s1 != ~0u ? (uint32_t)TWORD(c, s1 & 0xff, 0, 0) : 0u;
            const uint32_t f2 = s2 != ~0u ? (uint32_t)TWORD(c, s2 & 0xff, 0, 0) : 0u, f3 = s3 != ~0u ? (uint32_t)TWORD(c, s3 & 0xff, 0, 0) : 0u;
            // (a wake-up writes its own task's word only: the four words read together are what four reads in turn would have seen)
            if (s0 != ~0u) { TWORD(c, s0 & 0xff, 0, 0) = f0 | TF_KILLED; wake_with<K>(c, L, s0 & 0xff, (f0 >> 8) & 0xffff, f0 | TF_KILLED); }
            if (s1 != ~0u) { TWORD(c, s1 & 0xff, 0, 0) = f1 | TF_KILLED; wake_with<K>(c, L, s1 & 0xff, (f1 >> 8) & 0xffff, f1 | TF_KILLED); }
            if (s2 != ~0u) { TWORD(c, s2 & 0xff, 0, 0) = f2 | TF_KILLED; wake_with<K>(c, L, s2 & 0xff, (f2 >> 8) & 0xffff, f2 | TF_KILLED); }
            if (s3 != ~0u) { TWORD(c, s3 & 0xff, 0, 0) = f3 | TF_KILLED; wake_with<K>(c, L, s3 & 0xff, (f3 >> 8) & 0xffff, f3 | TF_KILLED); }
            if (!more) break;
            last = s3; any_last = true;
        }
        return;
    }
    uint32_t last = 0xffffffffu;                            // "none yet": sequence numbers are < 2^24
    for (;;) {
        uint32_t best = 0xffffffffu, best_seq = 0xffffffffu;
        for (uint32_t t = 0; t < c.P.max_tasks; t++) {
            if (K::G && !((AMASK(t >> 5) >> (t & 31)) & 1)) continue;      // (the alive mask, LDS: no read of a free slot's flag word)
            uint32_t f = TWORD(c, t, 0, 0);
            if (!(f & TF_ALIVE) || (PROGW(c, f >> 24) & 0xff) != node) continue;
            uint32_t sw = TWORD(c, t, 1, 1);
            uint32_t seq = sw & 0xffffff;
            if ((sw >> 24) != info_gen) continue;
            if ((last == 0xffffffffu || seq > last) && seq < best_seq) { best = t; best_seq = seq; }
        }
        if (best == 0xffffffffu) break;
        uint32_t f = TWORD(c, best, 0, 0);
        TWORD(c, best, 0, 0) = f | TF_KILLED;
        wake<K>(c, L, best, (f >> 8) & 0xffff);
        last = best_seq;
    }
}

template <class K>
__device__ __forceinline__ uint32_t task_finish(const Ctx& c, Lane& L, uint32_t slot, uint32_t outcome, bool guard, FinishKnown kn);

// node.paused.clear() (task/mod.rs:365,392): the parked Runnables of `node` are dropped, in order.
template <class K>
__device__ __forceinline__ void paused_clear(const Ctx& c, Lane& L, uint32_t node) {
    if (!c.P.uses_pause) return;
    uint32_t n = PAUSEW(0), keep = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t slot = PAUSEW(1 + i);
        if ((PROGW(c, TWORD(c, slot, 0, 0) >> 24) & 0xff) == node) task_finish<K>(c, L, slot, H_CANCELLED);
        else { PAUSEW(1 + keep) = slot; keep++; }
    }
    PAUSEW(0) = keep;
}

template <class K>
__device__ __forceinline__ void node_kill(const Ctx& c, Lane& L, uint32_t node) {        // TaskHandle::kill_id (task/mod.rs:362-371)
    paused_clear<K>(c, L, node);
    uint32_t g = NODE_INFO_GEN(node);
    NODEW(0) |= 1u << node;
    if (g == 0) NODEW(2) |= 1u << node;
    info_kill<K>(c, L, node, g);
    for (uint32_t i = 0; i < c.P.n_socks; i++)               // NetSim::reset_node (network.rs:142-147)
        if ((SOCKW(c, i) & 0xff) == node) {
            SW(c, i, 0) &= ~(1u | (0x7fu << 25));              // (and nobody's guard matters any more)
            // a socket whose Endpoint is already gone dies with its table entry: so do the connections queued there
            if (K::FC && c.P.uses_chan && SW(c, i, 1) == ~0u && (SW(c, i, 2 + c.P.mbox_regs + 2 * c.P.mbox_msgs) & 0xf)) sock_drop_acceptq<K>(c, L, i);
        }
}

template <class K>
__device__ __forceinline__ void node_restart(const Ctx& c, Lane& L, uint32_t node) {     // TaskHandle::restart (task/mod.rs:374-401)
    uint32_t g = NODE_INFO_GEN(node);
    if (g == 0) NODEW(2) |= 1u << node;
    uint32_t w = NODEW(4 + (node >> 2)), sh = (node & 3) * 8;
    NODEW(4 + (node >> 2)) = (w & ~(0xffu << sh)) | (((g + 1) & 0xff) << sh);      // new_info
    NODEW(0) &= ~(1u << node);
    NODEW(1) &= ~(1u << node);
    paused_clear<K>(c, L, node);
    info_kill<K>(c, L, node, g);                             // old_info.kill()
    for (uint32_t p = 1; p < c.P.n_progs; p++) {             // init(&Spawner { new info })
        uint32_t pw = PROGW(c, p);
        if ((pw & 0xff) == node && ((pw >> 8) & MADSIM_PROG_INIT)) spawn_task<K>(c, L, p, false);
    }
}

}  // namespace madsim_k

#endif
