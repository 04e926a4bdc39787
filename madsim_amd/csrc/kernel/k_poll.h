// k_poll.h — poll_task: one poll of a task program (the workload VM).
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_POLL_H
#define MADSIM_K_POLL_H

namespace madsim_k {

// TimeHandle::sleep_until (time/mod.rs:118-124): 1 ms floor
__device__ __forceinline__ uint64_t sleep_deadline(const Lane& L, uint64_t deadline) {
    uint64_t m = L.clock + NS_PER_MS;
    return deadline > m ? deadline : m;
}

// NetSim::rand_delay up to the creation of its Sleep (net/mod.rs:287-292): returns the Sleep's deadline.
template <class K>
__device__ __forceinline__ uint64_t rand_delay_deadline(const Ctx& c, Lane& L) {
    uint64_t delay = (uint64_t)gen_range_small<K, 5>(c, L) * 1000ull;
    if (c.P.buggify) {
        if (gen_bool_pint<K>(c, L, c.P.bug_pint, 0)) delay = (uint64_t)(1 + gen_range_small<K, 4>(c, L)) * NS_PER_S;
    }
    return sleep_deadline(L, L.clock + delay);
}

// madsim_config_t.loss_table[i & 3] as Bernoulli parameters.  A select chain, not P.loss_table_pint[i]: a per-lane index
// into the by-value kernel-argument block makes the compiler copy the whole block to scratch.
__device__ __forceinline__ uint64_t loss_pint_at(const KParams& P, uint32_t i) {
    i &= 3;
    return i == 0 ? P.loss_table_pint[0] : i == 1 ? P.loss_table_pint[1] : i == 2 ? P.loss_table_pint[2] : P.loss_table_pint[3];
}
__device__ __forceinline__ uint32_t loss_always_at(const KParams& P, uint32_t i) {
    i &= 3;
    return i == 0 ? P.loss_table_always[0] : i == 1 ? P.loss_table_always[1] : i == 2 ? P.loss_table_always[2] : P.loss_table_always[3];
}

// A registration word identifies its receiver by 8 bits of receive sequence number and 8 bits of task generation.  A dead
// registration (timed-out / dropped receive) equal to a NEW word — which would make it look live — can only exist once one
// of the two has wrapped: this task's rxseq (TF_RXWRAP) or its slot's generation (>= 256 instances).  Only then is the
// mailbox scanned for a twin (=> MADSIM_UNSUPPORTED — no limit lifts it; the oracle, whose sequence numbers do not wrap, checks the same
// agreement of the low bytes at the same registration: oracle/madsim_oracle.c reg_model_limits).
__device__ __forceinline__ bool may_have_twin(uint32_t flags, uint32_t gen) { return (flags & TF_RXWRAP) || gen > 0xff; }

// Instruction fetch.  Workloads with ephemeral Endpoints (full-address builds only): an Endpoint operand naming a handle
// is replaced by the candidate entry the handle's last bind took (k_state.h sock_resolve) — except MS_OP_BIND's own.
// (Destination operands never name a handle: validate().)
// The fetch is also where a handle that no longer names its socket is caught (k_state.h handle_names_its_socket): every fetch is
// followed by the instruction's execution — or its next poll — in the same task poll with no socket state changing in between
// (the callers fetch after the previous op's effects and never while the task is panicking), which is exactly when the oracle
// enters the op's case.  `drop(ep)` of such a name is a no-op, as in the oracle: the fetch hands back `jmp pc + 1`.
template <class K>
__device__ __forceinline__ uint4 insn_fetch(const Ctx& c, Lane& L, uint32_t pc) {
    uint4 in = INSN(c, pc);
    if (!PLAIN_ADDR && c.P.uses_eph) {
        const uint32_t op = in.x & 0xff;
        const bool own = op == MS_OP_SEND || op == MS_OP_CONNECT || op == MS_OP_RPC_CALL || op == MS_OP_REPLY || op == MS_OP_RECV ||
                         op == MS_OP_RECV_TIMEOUT || op == MS_OP_CLOSE || op == MS_OP_ACCEPT || op == MS_OP_RPC_REPLY;   // a names an Endpoint
        const uint32_t s = (in.x >> 8) & 0xff;
        if (own && (SOCKW(c, s) & 0x8000u)) {
            if (handle_names_its_socket<K>(c, s)) in.x = (in.x & ~0xff00u) | (sock_resolve<K>(c, s) << 8);
            else if (op == MS_OP_CLOSE) in = make_uint4(MS_OP_JMP | ((pc + 1) << 16), 0, 0, 0);
            else { OVF_SET(L, OVF_MODEL); in.x = (in.x & ~0xff00u) | (sock_resolve<K>(c, s) << 8); }
        }
    }
    return in;
}

__device__ __forceinline__ bool is_light(uint32_t op) {
    return op == MS_OP_ASSERT_VAL || op == MS_OP_DJNZ || op == MS_OP_SET || op == MS_OP_JMP || op == MS_OP_TRACE || op == MS_OP_JEQ;
}

// One poll of a task's future (Runnable::run, task/mod.rs:279-283).  `u0` is the task's unit0, held
// in registers for the whole poll and written back by the caller.  Returns true if the task panicked.
//
// A poll is a sequence of rounds; one round = [A] resolve the await the task is parked on and run its
// completion action, [B] run the cheap straight-line ops that follow, [C] begin the next awaiting op.
// In steady state every poll is exactly one round, and all lanes walk A -> B -> C together, so the
// expensive primitives (RNG draws, heap pushes, link test, mailbox scan) sit at fixed points that the
// whole wave reaches at the same time.
template <class K>
__device__ __forceinline__ bool poll_task(const Ctx& c, Lane& L, const uint32_t slot, uint4& u0, uint4 u1, PollPrefetch pp) {
    enum : uint32_t { ST_RUN = 0, ST_PENDING = 1, ST_FINISHED = 2, ST_PANIC = 3 };
    const KParams& P = c.P;
    // a full registration list: short of the ceiling of 255 a larger mbox_regs lifts it (a capacity verdict, the seed is run again); AT the ceiling the
    // 256th registration leaves the workload model — the oracle's Vec holds the same registrations, it says MADSIM_UNSUPPORTED at the same receive
    const uint32_t REGS_FULL = P.mbox_regs >= MADSIM_MAX_MBOX_REGS ? (uint32_t)OVF_MODEL : (uint32_t)OVF_CAP;
    bool u1_dirty = false;
    uint32_t pc = u0.y & 0xffff, sub = (u0.y >> 16) & 0xff, from = u0.y >> 24;
    const uint32_t gen = (u0.x >> 8) & 0xffff;
    const uint32_t node = PROGW(c, u0.x >> 24) & 0xff;
    uint32_t st = ST_RUN;

    // timeout(d, ep.recv_from(tag)) = select_biased! { fut, sleep } (time/mod.rs:128-140): poll the recv future,
    // then the timeout's Sleep — which registers ANOTHER timer on every not-elapsed poll (time/sleep.rs:51-53).
    // Returns true when the op completed (Ok or Err(Elapsed)); otherwise the task is Pending.
    // (`first_poll`: the call is the op's first poll, from [C] — every Sleep it registers is new; on later polls a Sleep that
    // was pending before registers its timer AGAIN: the `again` argument of timer_schedule, k_timer.h dedup_note)
    bool first_poll = false;
    // this task's connection unit: global-state builds hold it in registers for the poll (it came with unit 0: k_state.h
    // PollPrefetch) and write through; the other builds read LDS as before
    constexpr bool CU_FULL = Hoist<K>::CHAN;      // (else word 0 only: k_state.h poll_prefetch)
    auto cu_get = [&]() -> uint4 { if (CU_FULL) return pp.cu; return (uint4)TU(c, slot, c.P.chan_unit); };
    auto cu0_get = [&]() -> uint32_t { if (K::G) return pp.cu.x; return (uint32_t)TWORD(c, slot, c.P.chan_unit, 0); };
    auto cu0_set = [&](uint32_t v) { if (K::G) pp.cu.x = v; TWORD(c, slot, c.P.chan_unit, 0) = v; };
    auto cu_set = [&](const uint4& v) { if (CU_FULL) pp.cu = v; else if (K::G) pp.cu.x = v.x; TU(c, slot, c.P.chan_unit) = v; };
    // this task's rpc unit (k_state.h PollPrefetch rq0 / rq1): the poll's copy where it has one, written through.  (Macros, not lambdas: builds
    // without the copy must compile to what they were — the base-op kernels' register allocation follows every value the front end makes.)
#ifdef MADSIM_EMU          // (the host-compiled kernel checks the copies against memory at every use)
#define RQ_GET(k_) ((HoistRpc<K>::ON && ((k_) ? pp.rq1 : pp.rq0) != (uint32_t)TWORD(c, slot, P.rpc_unit, (k_)) ? (void)OVF_SET(L, OVF_BUG) : (void)0), (uint32_t)TWORD(c, slot, P.rpc_unit, (k_)))
#else
#define RQ_GET(k_) (HoistRpc<K>::ON ? ((k_) ? pp.rq1 : pp.rq0) : (uint32_t)TWORD(c, slot, P.rpc_unit, (k_)))
#endif
#define RQ_SET(k_, v_) do { const uint32_t rq_v_ = (v_); if (HoistRpc<K>::ON) { if (k_) pp.rq1 = rq_v_; else pp.rq0 = rq_v_; } TWORD(c, slot, P.rpc_unit, (k_)) = rq_v_; } while (0)
    // stage [A]'s requests for the ops that send in this round (HdrPrefetch): destination table entry, its header, the caller's own header (rpc call)
    uint32_t a_dst = ~0u, a_hdr = 0, a_own = 0;
    auto recv_timeout_poll = [&]() -> bool {
        bool fut_ready = false;
        bool d1_new = false;
        if (sub == 1 && (u0.x & TF_INBOX)) {                 // oneshot ready -> rand_delay (endpoint.rs:145)
            u0.x &= ~TF_INBOX;
            from = u0.y >> 24;
            if (K::FR && P.uses_rpc && (INSN(c, pc).x >> 24) >= MADSIM_TAG_RPC_FIRST) RQ_SET(0, RQ_GET(1));
            uint64_t d1 = rand_delay_deadline<K>(c, L);
            u1.z = (uint32_t)d1; u1.w = (uint32_t)(d1 >> 32); u1_dirty = true;
            sub = 2;
            d1_new = true;
        }
        if (sub == 2) {
            uint64_t d1 = u64of(u1.z, u1.w);
            if (L.clock >= d1) fut_ready = true;
            else timer_schedule<K>(c, L, d1, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true, !d1_new);
        }
        if (fut_ready) return true;                          // Ok((len, from))
        uint64_t d2;
        if (K::G) d2 = u64of(pp.d2lo, pp.d2hi);              // (came with unit 0: k_state.h PollPrefetch)
        else { uint4 u2 = TU(c, slot, 2); d2 = u64of(u2.z, u2.w); }
        if (L.clock >= d2) {                                 // Err(Elapsed): the recv future is dropped
            u1.x = (u1.x & ~0xffu) | (((u1.x & 0xff) + 1) & 0xff); u1_dirty = true;   // its oneshot::Receiver is gone
            if ((u1.x & 0xff) == 0) u0.x |= TF_RXWRAP;
            u0.x &= ~TF_INBOX;
            u0.w = MADSIM_VAL_TIMEOUT;
            return true;
        }
        timer_schedule<K>(c, L, d2, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true, !first_poll);
        st = ST_PENDING;
        return false;
    };

    // Endpoint::call / call_timeout (net/rpc.rs:96-131) from its first poll on.  sub 1: send_to_raw's rand_delay;
    // sub 2: recv_from_raw(rsp_tag)'s oneshot; sub 3: its rand_delay.  With a timeout, the timeout's Sleep is polled after
    // the call future on every poll and registers ANOTHER timer each time (select_biased!, time/sleep.rs:51-53).
    // Returns true when the op completed (Ok, Err(TimedOut)) or the task panicked (st).
    auto rpc_call_poll = [&]() -> bool {
        const uint4 ci = insn_fetch<K>(c, L, pc);
        const uint32_t ca = (ci.x >> 8) & 0xff, cb = ci.x >> 16, cimm = ci.y;
        const uint32_t dst = cb & 0xff;
        if (sub == 1) {
            uint64_t d1 = u64of(u1.z, u1.w);
            if (L.clock < d1) {
                timer_schedule<K>(c, L, d1, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true, !first_poll);
            } else {
                // the caller's pending receive doubles as the rsp_tag: registration word >> 8 (see mailbox_deliver)
                const uint32_t rxseq = ((u1.x & 0xff) + 1) & 0xff;
                const uint32_t reg = 0xffu | (slot << 8) | (rxseq << 16) | ((gen & 0xff) << 24);
                uint64_t lat; int ds;
                // hooks_req.get(&node): `if !hook(&msg) { return Ok(()) }` before try_send (net/mod.rs:307-311): no draws
                bool hooked = false;
                if (P.uses_hooks) {
                    const uint32_t hw = HOOKW(SOCKW(c, ca) & 0xff);
                    hooked = (hw & 1) && ((hw >> 10) & 0xff) == (cb >> 8) && ((hw & 2) || ((hw >> 2) & 0xff) == (cimm & 0xff));
                }
                uint32_t lb = 0, to_idx = dst, to_addr = SOCKW(c, dst);
                if (!hooked) ipvs_rewrite<K>(c, to_idx, to_addr);          // after the hook, before try_send (net/mod.rs:312-317)
                uint32_t dh = 0;
                // (global-state builds: this Endpoint's header, which the registration below wants, requested with net_try_send's reads)
                // (`if constexpr`: a build without the requests must not even capture them — the closure's shape moves the base-op builds' registers)
                uint32_t h_own, kn_idx = ~0u, kn_hdr = 0;
                if constexpr (HdrPrefetch<K>::ON) {
#ifdef MADSIM_EMU
                    if (a_dst != to_idx || a_own != (uint32_t)SW(c, ca, 0)) OVF_SET(L, OVF_BUG);     // stage [A] requested exactly these
#endif
                    h_own = a_own; kn_idx = a_dst; kn_hdr = a_hdr;
                } else h_own = K::G ? (uint32_t)SW(c, ca, 0) : 0u;
                const int sent = hooked ? 0 : net_try_send<K>(c, L, SOCKW(c, ca) & 0xff, to_addr, to_idx, &lat, &ds, &lb, &dh, kn_idx, kn_hdr);
                if (sent < 0) { st = ST_PANIC; return true; }
                if (sent) {
                    uint32_t sgen = (dh >> 1) & 0xff;
                    uint32_t meta = (EV_DELIVER << EV_SHIFT) | (sgen << 21) | ((cb >> 8) << 13) | ((ca | (lb << 6)) << 6) | (uint32_t)ds;
                    timer_schedule<K>(c, L, L.clock + lat, meta, (cimm & 0xff) | (reg & 0xffffff00u), false);
                }
                // recv_from_raw(rsp_tag): Mailbox::recv (endpoint.rs:353-362); no queued message can carry a fresh tag
                u1.x = (u1.x & ~0xffu) | rxseq; u1_dirty = true;
                if (rxseq == 0) u0.x |= TF_RXWRAP;
                u0.x &= ~TF_INBOX;
                uint32_t h = K::G ? h_own : (uint32_t)SW(c, ca, 0);
                uint32_t nreg = (h >> 9) & 0xff;
                if (may_have_twin(u0.x, gen)) for (uint32_t i = 0; i < nreg; i++) if (SW(c, ca, 2 + i) == reg) OVF_SET(L, OVF_MODEL);   // 8-bit rxseq wrapped onto a dead twin
                if (nreg >= P.mbox_regs) OVF_SET(L, REGS_FULL);
                else {
                    SW(c, ca, 2 + nreg) = reg;
                    SW(c, ca, 0) = (h & ~(0xffu << 9)) | ((nreg + 1) << 9);
                }
                sub = 2;
            }
        }
        bool d1_new = false;
        if (sub == 2 && (u0.x & TF_INBOX)) {                 // oneshot ready -> rand_delay (endpoint.rs:145)
            u0.x &= ~TF_INBOX;
            from = u0.y >> 24;
            uint64_t d1 = rand_delay_deadline<K>(c, L);
            u1.z = (uint32_t)d1; u1.w = (uint32_t)(d1 >> 32); u1_dirty = true;
            sub = 3;
            d1_new = true;
        }
        if (sub == 3) {
            uint64_t d1 = u64of(u1.z, u1.w);
            if (L.clock >= d1) {
                if (PLAIN_ADDR ? from != dst : !addr_eq(addr_of_from(c, from), SOCKW(c, dst))) st = ST_PANIC;   // assert_eq!(from, dst) rpc.rs:126
                return true;
            }
            timer_schedule<K>(c, L, d1, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true, !d1_new);
        }
        if (cimm >> 8) {
            uint64_t d2;
            if (K::G) d2 = u64of(pp.d2lo, pp.d2hi);
            else { uint4 u2 = TU(c, slot, 2); d2 = u64of(u2.z, u2.w); }
            if (L.clock >= d2) {                             // Err(Elapsed) -> TimedOut: the call future is dropped
                if (sub >= 2) { u1.x = (u1.x & ~0xffu) | (((u1.x & 0xff) + 1) & 0xff); u1_dirty = true; u0.x &= ~TF_INBOX; if ((u1.x & 0xff) == 0) u0.x |= TF_RXWRAP; }
                u0.w = MADSIM_VAL_TIMEOUT;
                return true;
            }
            timer_schedule<K>(c, L, d2, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true, !first_poll);
        }
        st = ST_PENDING;
        return false;
    };

    // accept1's conn_rx.recv() (endpoint.rs:200): take the oldest queued connection or park. true = op completed.
    auto accept_check = [&](uint32_t a) -> bool {
        uint32_t base = 2 + P.mbox_regs + 2 * P.mbox_msgs;
        const uint32_t q_lo = SW(c, a, base);
        const uint32_t q_hi = Hoist<K>::CHAN ? (uint32_t)SW(c, a, base + 2) : 0u;      // (global-state builds: both halves of the queue word at once,
        uint32_t ha = Hoist<K>::CHAN ? (uint32_t)SW(c, a, 0) : 0u;                     //  and the header guard_acquire wants below)
        uint32_t n = q_lo & 0xf;
        if (n == 0) { SW(c, a, base + 1) = 1u | (slot << 1) | (gen << 9); st = ST_PENDING; return false; }
        uint64_t q = Hoist<K>::CHAN ? u64of(q_lo, q_hi) : acceptq_load<K>(c, a);
        uint32_t id = (uint32_t)(q >> 4) & 0x7f;
        acceptq_store<K>(c, a, (uint64_t)(n - 1) | ((q >> 11) << 4));           // pop front
        uint32_t cx = cu0_get();
        const uint32_t cw_p = Hoist<K>::CHAN ? (uint32_t)CONNW(id, 0) : 0u;           // (with the drop's reads below, if any)
        if ((cx & 0xff) != 0xff) {
            conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1, (u0.x & TF_KILLED) != 0);
            if (Hoist<K>::CHAN) ha = SW(c, a, 0);                                      // (the drop may have released a guard of `a`)
        }
        cu0_set(id | (1u << 8));                                           // server side
        // Sender { _guard: self.guard.clone(), tx }, Receiver { _guard: self.guard.clone(), rx } (endpoint.rs:203-210)
        // (the header word read before the drop stands when the drop was of ANOTHER connection; a task that accepts its own
        //  outgoing connection — bind a; connect1(a, a); accept1(a) — has just closed this one's client handles: read it again)
        const bool own = (cx & 0xff) == id;
        CONNW(id, 0) = ((Hoist<K>::CHAN && !own ? cw_p : (uint32_t)CONNW(id, 0)) & ~(0x7fu << 25)) | (a << 25) | (1u << 31);
        if (Hoist<K>::CHAN) guard_acquire_with<K>(c, L, a, ha); else guard_acquire<K>(c, L, a);
        return true;
    };
    // the receiver stream of channel() (net/mod.rs:386-400) from "a payload is in hand" (sub 1): either sleep(backoff)
    // while its State is None (sub 2) or sleep_until(arrive_time) (sub 3).  Always ends Pending (1 ms floor).
    auto crecv_arm = [&](const uint4& u3) {
        uint64_t arrive = u64of(u3.z, u3.w), d;
        if (arrive != ~0ull) { d = sleep_deadline(L, arrive); sub = 3; }
        else { d = sleep_deadline(L, L.clock + (uint64_t)(u3.x >> 16) * NS_PER_MS); sub = 2; }
        u1.z = (uint32_t)d; u1.w = (uint32_t)(d >> 32); u1_dirty = true;
        timer_schedule<K>(c, L, d, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true);
        st = ST_PENDING;
    };

    HeapPre heap_pre = {~0u, make_uint2(0, 0)};          // (k_timer.h timer_push_prefetch: requested when a round begins, used by the flush that ends it)
    for (;;) {
        // global-state builds: the Timer::add calls of the previous round happen here, at one site for the whole wave
        timer_flush<K>(c, L, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, heap_pre);
        PROBE_FLUSH();          // (EXP_PROF builds: the pushes apart from the wait for the other lanes' further rounds behind the loop)
        if (st != ST_RUN) break;
        if (PushPrefetch<K>::ON) heap_pre = timer_push_prefetch<K>(c, L);
        // (pc < n_insns always: validate() checks jump targets and that the table ends in DONE / JMP / PANIC)
        uint4 in = insn_fetch<K>(c, L, pc);
        uint32_t op = in.x & 0xff, a = (in.x >> 8) & 0xff, b = in.x >> 16, imm = in.y;

        PROBE(5);
        REG(2);
        // Requests of stage [A]: the ops whose rand_delay has elapsed send in this round — send_to / reply / rpc_reply in the Sleep branch below, the
        // call of an rpc in rpc_call_poll — and each used to wait for its destination's header inside its own divergent block.  The lanes of all of
        // them request it here, together (and the rpc caller its own Endpoint's header, which the registration wants).
        if constexpr (HdrPrefetch<K>::ON) a_dst = ~0u;
        if constexpr (HdrPrefetch<K>::ON) if (sub != 0 && sub < SUB_JOIN_WAIT && L.clock >= u64of(u1.z, u1.w)) {
            const bool snd = op == MS_OP_SEND || op == MS_OP_REPLY || (K::FR && op == MS_OP_RPC_REPLY);
            const bool call = K::FR && op == MS_OP_RPC_CALL && sub == 1;
            if (snd || call) {
                a_dst = (op == MS_OP_SEND || call) ? (b & 0xff) : (from & 0x3f);
                a_hdr = SW(c, a_dst, 0);
                if (call) a_own = SW(c, a, 0);
            }
        }
        // ================= [A] the task is parked on an await of this op =========================
        if (sub != 0 && sub < SUB_JOIN_WAIT) {
            bool completed = false;                        // this op is done: step to the next one below
            if (op == MS_OP_RECV && sub == 1) {            // oneshot::Receiver (endpoint.rs:142-144)
                REG(4);
                if (!(u0.x & TF_INBOX)) st = ST_PENDING;       // (no `break`: leaves at the check behind [B], like every Pending)
                else {
                    u0.x &= ~TF_INBOX;
                    from = u0.y >> 24;
                    if (K::FR && P.uses_rpc && (b >> 8) >= MADSIM_TAG_RPC_FIRST)   // (rsp_tag, req, data) = *data.downcast()
                        RQ_SET(0, RQ_GET(1));
                    sub = 2;                               // -> rand_delay, begun in [C]
                }
            } else if (op == MS_OP_YIELD) {
                completed = true;
            } else if (K::FT && op == MS_OP_RECV_TIMEOUT) {
                completed = recv_timeout_poll();            // (false: st is Pending — the round leaves behind [B], no `break` here)
            } else if (K::FR && op == MS_OP_RPC_CALL) {
                completed = rpc_call_poll() && st == ST_RUN;
            } else if (K::FC && op == MS_OP_ACCEPT && sub == 2) {
                completed = accept_check(a);
            } else {                                       // a Sleep (time/sleep.rs:47-54)
                uint64_t deadline = u64of(u1.z, u1.w);
                REG(3);
                // (base-op builds keep no deadline: there a Sleep is only polled again once its own timer has fired)
                if (K::LIFE && L.clock < deadline) {       // not elapsed: register ANOTHER timer
                    REG(5);
                    timer_schedule<K>(c, L, deadline, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true, true);
                    st = ST_PENDING;
                }
                else if (K::FC && op == MS_OP_ACCEPT) {    // rand_delay done -> conn_rx.recv()
                    sub = 2;
                    accept_check(a);                       // (false: st is Pending)
                } else if (K::FC && op == MS_OP_CRECV) {
                    uint4 u3 = cu_get();
                    if (sub == 2) {                        // sleep(backoff) done: backoff = min(2 * backoff, 10 s); retry the link
                        uint32_t bo = (u3.x >> 16) * 2; if (bo > 10000) bo = 10000;
                        uint32_t cw = CONNW(u3.x & 0xff, 0);
                        uint64_t arrive = chan_test_link<K>(c, L, cw, 1 - ((u3.x >> 8) & 1));
                        if (arrive == CHAN_LINK_PANIC) st = ST_PANIC;
                        else {
                            u3.x = (u3.x & 0xffff) | (bo << 16); u3.z = (uint32_t)arrive; u3.w = (uint32_t)(arrive >> 32);
                            cu_set(u3);
                            crecv_arm(u3);                 // (always ends Pending)
                        }
                    }
                    else u0.w = u3.y;                      // sub 3: sleep_until(arrive_time) done -> yield value
                } else if (K::FC && op == MS_OP_CONNECT) {   // NetSim::connect1 (net/mod.rs:345-363)
                    uint32_t cx = cu0_get();
                    if ((cx & 0xff) != 0xff) { conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1, (u0.x & TF_KILLED) != 0); cu0_set(cx | 0xff); }
                    uint64_t lat; int ds; uint32_t lb;
                    uint32_t dial = b & 0xff, dial_addr = SOCKW(c, dial);
                    ipvs_rewrite<K>(c, dial, dial_addr);          // channel() below is built from the rewritten dst (net/mod.rs:345-357)
                    uint32_t dh;
                    // (global-state builds, plain addresses — the listener's socket IS table entry `dial`: its parked acceptor's word goes out with
                    //  net_try_send's header request, so the acceptor's flag word can go out with the batch below instead of after it)
                    const bool acc_early = Hoist<K>::CHAN && PLAIN_ADDR;
                    const uint32_t acc_e = acc_early ? (uint32_t)SW(c, dial, 3 + P.mbox_regs + 2 * P.mbox_msgs) : 0u;
                    const int sent = net_try_send<K>(c, L, SOCKW(c, a) & 0xff, dial_addr, dial, &lat, &ds, &lb, &dh);
                    if (sent < 0) st = ST_PANIC;
                    else if (!sent) {
                        u0.w = MADSIM_VAL_REFUSED;
                    } else {
                        uint32_t id = 0;
                        uint32_t base = 2 + P.mbox_regs + 2 * P.mbox_msgs;
                        uint64_t q;
                        uint32_t own_p = 0, acc_p = 0, ha_p = 0, awf_p = 0;
                        if (Hoist<K>::CHAN) {
                            // one round trip for everything the rest of the op reads: the first four connection headers (the free-slot
                            // search), the listener's queue word, owner word and parked acceptor, this Endpoint's header.  What is
                            // stored between here and their use — connection words, the guard count of `a` — touches none of them.
                            const uint32_t c0 = CONNW(0, 0), c1 = P.max_conns > 1 ? (uint32_t)CONNW(1, 0) : 1u,
                                           c2 = P.max_conns > 2 ? (uint32_t)CONNW(2, 0) : 1u, c3 = P.max_conns > 3 ? (uint32_t)CONNW(3, 0) : 1u;
                            const uint32_t q_lo = SW(c, ds, base), q_hi = SW(c, ds, base + 2);
                            own_p = SW(c, ds, 1); acc_p = acc_early ? acc_e : (uint32_t)SW(c, ds, base + 1); ha_p = SW(c, a, 0);
                            if (acc_early && (acc_e & 1)) awf_p = TWORD(c, (acc_e >> 1) & 0xff, 0, 0);
                            q = u64of(q_lo, q_hi);
                            id = !(c0 & 1) ? 0u : !(c1 & 1) ? 1u : !(c2 & 1) ? 2u : !(c3 & 1) ? 3u : 4u;
                            while (id >= 4 && id < P.max_conns && (CONNW(id, 0) & 1)) id++;      // (beyond the first four: one at a time)
                        } else {
                            while (id < P.max_conns && (CONNW(id, 0) & 1)) id++;
                            q = acceptq_load<K>(c, (uint32_t)ds);
                        }
                        // (below the ceiling of 127 connections a larger max_conns lifts it; AT the ceiling the 128th leaves the model: the oracle
                        //  fills its connection Vec lowest free slot first like this and says MADSIM_UNSUPPORTED at the same connect1)
                        if (id >= P.max_conns) { OVF_SET(L, P.max_conns >= MADSIM_MAX_CONNS ? OVF_MODEL : OVF_CAP); }
                        else {
                            CONNW(id, 0) = 1u | (a << 1) | (dial << 7) | (0xfu << 13);     // client Endpoint, the address it dialled
                            CONNW(id, 1) = 0; CONNW(id, 2) = 0;
                            cu0_set(id);                                        // client side
                            if (Hoist<K>::CHAN) guard_acquire_with<K>(c, L, a, ha_p);
                            else guard_acquire<K>(c, L, a);        // Sender / Receiver { _guard: self.guard.clone(), .. } (endpoint.rs:181-190)
                            u0.w = 0;
                            if ((Hoist<K>::CHAN ? own_p : (uint32_t)SW(c, ds, 1)) == ~0u) {   // the listener's Endpoint is gone (connections it accepted hold the
                                conn_drop_raw<K>(c, L, id, 1);     // address): `let _ = conn_tx.try_send(..)` drops (tx2, rx1) here
                            } else {
                                uint32_t n = (uint32_t)q & 0xf;    // socket.new_connection -> conn_tx.try_send
                                // (conn_tx is unbounded; this queue word holds MADSIM_ACCEPTQ ids.  A ninth connection waiting for accept1 is
                                //  outside the model — no limit of madsim_limits_t grows the word — so it is MADSIM_UNSUPPORTED, decided at the
                                //  same instruction by the oracle, not a capacity verdict that a re-run could never resolve)
                                // (the queue word stays as it is: the seed's state is spoiled from here on, but never out of bounds —
                                //  OR-ing a ninth id over slot 7 made ids up to 127 that a later accept1 would index the connection table with)
                                if (n >= MADSIM_ACCEPTQ) OVF_SET(L, OVF_MODEL);
                                else {
                                    acceptq_store<K>(c, (uint32_t)ds, (q & ~0xfull) | (n + 1) | ((uint64_t)id << (4 + 7 * n)));
                                    uint32_t acc = Hoist<K>::CHAN ? acc_p : (uint32_t)SW(c, ds, base + 1);
                                    if (acc & 1) {
                                        SW(c, ds, base + 1) = 0;
                                        if (acc_early) {
#ifdef MADSIM_EMU
                                            if ((uint32_t)ds != dial || awf_p != (uint32_t)TWORD(c, (acc >> 1) & 0xff, 0, 0)) OVF_SET(L, OVF_BUG);
#endif
                                            wake_with<K>(c, L, (acc >> 1) & 0xff, acc >> 9, awf_p);
                                        } else wake<K>(c, L, (acc >> 1) & 0xff, acc >> 9);
                                    }
                                }
                            }
                        }
                    }
                } else if (op == MS_OP_BIND) {                    // Network::bind (network.rs:206-251)
                    // a specified, non-loopback IP must be the node's own (:215-222); table entries are per node for every
                    // kind, so another node's entry is "not available" too; then the (address, protocol) key must be free (:238-246)
                    uint32_t sw = SOCKW(c, a);
                    uint32_t bind_err = 0;
                    if ((sw & 0xff) != node) bind_err = MADSIM_VAL_ADDR_NOT_AVAILABLE;
                    else if (!PLAIN_ADDR && (sw & 0x8000u)) {
                        // port 0: the lowest port from 1 up that no socket of the node holds for this IP (:224-236) = the
                        // first free candidate entry of the handle; none left means the caller kept more Endpoints of this
                        // handle alive than the table has entries for (a capacity verdict, not an error of the workload)
                        const uint32_t base = (sw >> 16) & 0xff, nk = sw >> 24;
                        // An entry names ONE Endpoint at a time.  Bound again while the Endpoint of its previous bind is alive (in
                        // Rust: a second `Endpoint::bind("0.0.0.0:0")` beside the first) the two would have to coexist under one
                        // name: outside the model — the verdict says so, the oracle says the same (include/madsim_hip.h).  The
                        // handle word keeps {candidate, valid, the candidate's socket gen after that bind} to know.
                        if (handle_names_its_socket<K>(c, a) && SW(c, base + ((uint32_t)SW(c, a, 0) >> 25), 1) != ~0u) OVF_SET(L, OVF_MODEL);
                        uint32_t p = 0;
                        while (p < nk && find_exact<K>(c, node, (sw & 0x7fffu) | ((p + 1) << 16)) >= 0) p++;
                        // every candidate port of this (node, IP) is held — by addresses that outlived two Endpoints of one entry through their
                        // connections, say: no madsim_limits_t field adds table entries, so this is the model's edge, not a capacity (the oracle,
                        // which searches ports 1 .. 65 535, reports the same verdict when the port it finds lies beyond the candidates)
                        if (p == nk) { OVF_SET(L, OVF_MODEL); p = 0; }
                        SW(c, a, 0) = (p << 25) | (1u << 24) | (((((uint32_t)SW(c, base + p, 0) >> 1) + 1) & 0xff) << 16);
                        a = base + p;
                    }
                    else if ((PLAIN_ADDR ? find_bound<K>(c, a) : find_exact<K>(c, node, sw)) >= 0) bind_err = MADSIM_VAL_ADDR_IN_USE;
                    if (bind_err) {
                        if (!(b & 1)) st = ST_PANIC;                        // .unwrap()  (leaves below: `completed` stays false)
                        u0.w = bind_err;
                    } else {
                        if (b & 1) u0.w = 0;
                        if (b & 2) u0.w = SOCKW(c, a) >> 16;                // ep.local_addr().unwrap().port()
                        sock_bind<K>(c, a, slot, gen);                     // bound, gen+1, empty mailbox, owned by this task
                        u0.x |= TF_OWNER;
                        if (K::G) OMASK(a >> 5) |= 1u << (a & 31);
                        if (K::FC && P.uses_chan) { acceptq_store<K>(c, a, 0); SW(c, a, 3 + P.mbox_regs + 2 * P.mbox_msgs) = 0; }
                    }
                } else if (op == MS_OP_SEND || op == MS_OP_REPLY || (K::FR && op == MS_OP_RPC_REPLY)) {   // net/mod.rs:307-331
                    REG(6);
                    // the destination: the operand's table entry, or the address the request came from (`from`)
                    uint32_t dst = (op == MS_OP_SEND) ? (b & 0xff) : (from & 0x3f);
                    uint32_t dst_addr = (op == MS_OP_SEND || PLAIN_ADDR) ? SOCKW(c, dst) : addr_of_from(c, from);
                    ipvs_rewrite<K>(c, dst, dst_addr);
                    // (the request's rsp_tag word: requested here, looked at behind net_try_send — its destination-header request goes out in the
                    //  same round trip instead of waiting for this one inside the divergent block)
                    uint32_t rsp_tag_w = 0;
                    const bool is_rpc_reply = K::FR && op == MS_OP_RPC_REPLY;
                    if (is_rpc_reply) rsp_tag_w = RQ_GET(0);
                    // Network::try_send -> resolve_dest_node, test_link, socket lookup (network.rs:261-313)
                    uint64_t lat; int ds; uint32_t lb;
                    uint32_t dh = 0;
                    const int sent = net_try_send<K>(c, L, SOCKW(c, a) & 0xff, dst_addr, dst, &lat, &ds, &lb, &dh, a_dst, a_hdr);
                    if (is_rpc_reply) {                      // send_to_raw(from, rsp_tag, rsp): rpc.rs:172-175
                        b = 0xff00;
                        imm = (imm & 0xff) | (rsp_tag_w << 8);
                    }
                    if (sent < 0) st = ST_PANIC;
                    else if (sent) {
                        {
                            uint32_t sgen = (dh >> 1) & 0xff;
                            uint2 ev = ev_deliver_meta<K>(sgen, b >> 8, a | (lb << 6), (uint32_t)ds, imm, pc);
                            if (K::FR && P.uses_hooks && op == MS_OP_RPC_REPLY) {
                                // hooks_rsp.get(&dst_node) is cloned now and judges the message when the timer fires
                                // (net/mod.rs:321-328): the verdict is already fixed, so a dropped response is a timer
                                // that fires and delivers nothing
                                const uint32_t hw = HOOKW(SOCKW(c, (uint32_t)ds) & 0xff);
                                if ((hw & (1u << 18)) && ((hw & (1u << 19)) || ((hw >> 20) & 0xff) == (imm & 0xff))) ev = make_uint2(EV_NOP << EV_SHIFT, 0);
                            }
                            timer_schedule<K>(c, L, L.clock + lat, ev.x, ev.y, false);
                        }
                    }
                }
                completed = st == ST_RUN;                  // (a panic above leaves at the check behind [B]: no `break` in this chain)
            }
            if (completed) {                               // fall through to [B]/[C] with the next op: one pass per poll
                REG(12);
                sub = 0;
                // fused post-chain of this op (geometry.h build_tables): assert_eq!(val, ..), then djnz / jmp
                const uint32_t pf = in.w;
                if ((pf & 1) && u0.w != in.z) st = ST_PANIC;   // (no `break`: `op` stays this awaiting op, so [B] is skipped too)
                else {
                    pc = (pf >> 4) & 0x3fff;
                    if (pf & 2) {
                        uint32_t sh = ((pf >> 2) & 1) * 16;
                        uint32_t v = (((u0.z >> sh) & 0xffff) - 1) & 0xffff;
                        u0.z = (u0.z & ~(0xffffu << sh)) | (v << sh);
                        if (v) pc = pf >> 18;
                    } else if (pf & 8) {
                        pc = pf >> 18;
                    }
                    in = insn_fetch<K>(c, L, pc);
                    op = in.x & 0xff; a = (in.x >> 8) & 0xff; b = in.x >> 16; imm = in.y;
                }
            }
        }

        if constexpr (HdrPrefetch<K>::ON) a_dst = ~0u;     // (stage [A]'s requests are spent)
        PROBE(6);
        // ================= [B] cheap ops that never await ========================================
        while (is_light(op)) {
            REG(13);
            if (op == MS_OP_ASSERT_VAL) {
                if (u0.w != imm) { st = ST_PANIC; break; }
                pc++;
            } else if (op == MS_OP_DJNZ) {
                uint32_t sh = (a & 1) * 16;
                uint32_t v = (((u0.z >> sh) & 0xffff) - 1) & 0xffff;
                u0.z = (u0.z & ~(0xffffu << sh)) | (v << sh);
                pc = v ? b : pc + 1;
            } else if (op == MS_OP_SET) {
                u0.z = (a & 1) ? ((u0.z & 0xffffu) | (imm << 16)) : ((u0.z & 0xffff0000u) | (imm & 0xffffu));
                pc++;
            } else if (op == MS_OP_JMP) {
                pc = b;
            } else if (op == MS_OP_JEQ) {
                pc = (u0.w == imm) ? b : pc + 1;
            } else {                                       // MS_OP_TRACE
                uint64_t v = imm;
                if (b & 1) v += (u0.z >> ((a & 1) * 16)) & 0xffff;
                L.obs_hash = (L.obs_hash ^ v) * FNV_PRIME;
                pc++;
            }

            in = insn_fetch<K>(c, L, pc);
            op = in.x & 0xff; a = (in.x >> 8) & 0xff; b = in.x >> 16; imm = in.y;
        }
        if (st != ST_RUN) { if (K::G) continue; else break; }      // (global-state builds leave through the flush at the head)

        PROBE(7);
        REG(9);
        // ================= [C] begin the next op =================================================
        bool want_delay = false, want_sleep = false;
        uint64_t deadline = 0;
        // Every-class global-state builds: a receive that begins here — recv_from or timeout(recv_from) — wants its Endpoint's header and first queued message.
        // Both kinds request them HERE, in one instruction stream for the lanes of either, so the wave waits once for the two handlers, not once in
        // each (a wave runs every handler some lane is in; each handler's wait holds all 64 lanes).  Nothing is stored between here and the handlers.
        uint32_t c_hdr = 0, c_q0 = 0, c_q1 = 0;
        if (RecvPrefetch<K>::ON && ((op == MS_OP_RECV && sub == 0) || (K::FT && op == MS_OP_RECV_TIMEOUT))) {
            c_hdr = SW(c, a, 0); c_q0 = SW(c, a, 2 + P.mbox_regs); c_q1 = SW(c, a, 3 + P.mbox_regs);
        }
        // (... and the first reads of `spawn` and of a task that ends — the two rare-op handlers every pass of the topology runs: with the receives'
        //  requests above the whole stage waits once.  The free slot is the one spawn_task would find: the alive mask changes in spawn / finish only.)
        uint32_t g_slot = ~0u, g_old = 0, g_gen = 0, g_seq = 0, g_killed = 0, g_hw = 0;
        if constexpr (SwitchPrefetch<K>::ON) {
            if (op == MS_OP_SPAWN) {
                uint32_t sl = P.max_tasks;
                for (uint32_t wi = 0; wi < (P.max_tasks + 31) / 32 && sl == P.max_tasks; wi++) {
                    const uint32_t free_bits = ~AMASK(wi);
                    if (free_bits) sl = wi * 32 + (uint32_t)__builtin_ctz(free_bits);
                }
                if (sl < P.max_tasks) {
                    const uint32_t nd = PROGW(c, a) & 0xff;
                    g_slot = sl; g_old = TWORD(c, sl, 0, 0);
                    if (K::FN) { g_gen = NODEW(4 + (nd >> 2)); g_seq = NODEW(3); if (nd != node) g_killed = NODEW(2); }
                }
            } else if (op == MS_OP_DONE) g_hw = HW(u0.x >> 24);
        }
        if (op == MS_OP_RECV) {
            if (sub == 0) {                                // Mailbox::recv (endpoint.rs:353-362)
                REG(14);
                uint32_t tag = b >> 8;
                uint32_t rxseq = ((u1.x & 0xff) + 1) & 0xff;
                u1.x = (u1.x & ~0xffu) | rxseq; u1_dirty = true;
                if (rxseq == 0) u0.x |= TF_RXWRAP;
                u0.x &= ~TF_INBOX;
                uint32_t h = RecvPrefetch<K>::ON ? c_hdr : (uint32_t)SW(c, a, 0);
                uint32_t nreg = (h >> 9) & 0xff, nmsg = HDR_NMSG(h);
                uint32_t idx = 0, mbase = 2 + P.mbox_regs;
                while (idx < nmsg && ((RecvPrefetch<K>::ON && idx == 0 ? c_q0 : (uint32_t)SW(c, a, mbase + 2 * idx)) & 0xff) != tag) idx++;
                if (idx < nmsg) {
                    uint32_t m0, m1;
                    if (RecvPrefetch<K>::ON && idx == 0) { m0 = c_q0; m1 = c_q1; } else { m0 = SW(c, a, mbase + 2 * idx); m1 = SW(c, a, mbase + 2 * idx + 1); }
                    nmsg--;
                    if (!RecvPrefetch<K>::ON || idx != nmsg) {    // swap_remove (with the requests above: the last entry onto itself moves nothing)
                        SW(c, a, mbase + 2 * idx) = SW(c, a, mbase + 2 * nmsg);
                        SW(c, a, mbase + 2 * idx + 1) = SW(c, a, mbase + 2 * nmsg + 1);
                    }
                    u0.w = m1;
                    if (K::FR && P.uses_rpc && tag >= MADSIM_TAG_RPC_FIRST) { u0.w = m1 & 0xff; RQ_SET(0, m1 >> 8); }
                    from = (m0 >> 8) & 0xff;
                    sub = 2;                               // oneshot already holds the value
                    SW(c, a, 0) = HDR_SET_NMSG(h, nmsg);
                } else {
                    if (nreg >= P.mbox_regs) OVF_SET(L, REGS_FULL);          // (a runner verdict: the state no longer matters)
                    else {
                        const uint32_t reg = tag | (slot << 8) | (rxseq << 16) | ((gen & 0xff) << 24);
                        // dead registrations (timed-out / dropped receives, extended ops only) stay in the Vec like the
                        // reference's; an 8-bit rxseq that wraps onto one of them would make it look live: overflow, never a different answer
                        if ((K::FT || K::FN) && may_have_twin(u0.x, gen)) for (uint32_t i = 0; i < nreg; i++) if (SW(c, a, 2 + i) == reg) OVF_SET(L, OVF_MODEL);
                        SW(c, a, 2 + nreg) = reg;
                        SW(c, a, 0) = (h & ~(0xffu << 9)) | ((nreg + 1) << 9);
                        sub = 1;
                    }
                    st = ST_PENDING;
                }
            }
            want_delay = (sub == 2);                       // endpoint.rs:145 rand_delay
        } else if (op == MS_OP_SEND || op == MS_OP_REPLY || op == MS_OP_BIND || (K::FC && (op == MS_OP_CONNECT || op == MS_OP_ACCEPT)) || (K::FR && op == MS_OP_RPC_REPLY)) {
            want_delay = true;                             // net/mod.rs:306,344,457, endpoint.rs:198: rand_delay first
        } else if (K::FT && op == MS_OP_RECV_TIMEOUT) {
            uint32_t tag = b >> 8;
            uint64_t d2 = sleep_deadline(L, L.clock + (uint64_t)(b & 0xff) * NS_PER_S + imm);   // timeout()'s Sleep
            if (K::G) { buf_store64(c.gs, gs_addr_task(c, slot, 2 * 16u + 8u), make_uint2((uint32_t)d2, (uint32_t)(d2 >> 32))); pp.d2lo = (uint32_t)d2; pp.d2hi = (uint32_t)(d2 >> 32); }
            else { uint4 u2 = TU(c, slot, 2); u2.z = (uint32_t)d2; u2.w = (uint32_t)(d2 >> 32); TU(c, slot, 2) = u2; }
            uint32_t rxseq = ((u1.x & 0xff) + 1) & 0xff;       // Mailbox::recv (endpoint.rs:353-362)
            u1.x = (u1.x & ~0xffu) | rxseq; u1_dirty = true;
            if (rxseq == 0) u0.x |= TF_RXWRAP;
            u0.x &= ~TF_INBOX;
            const uint32_t mbase = 2 + P.mbox_regs;
            // Global-state builds: the first queued message is requested with the header (above, with recv_from's).  Three of four timeout(recv) calls
            // of the topology find their datagram queued already, and the scan, the message's words and the swap_remove's last entry — the same words,
            // when one message is queued — used to be four dependent round trips; now the header's, and one more only when something has to move.
            // (the other global-state builds request the three words here, in the handler)
            uint32_t h, q0 = c_q0, q1 = c_q1;
            if (RecvPrefetch<K>::ON) h = c_hdr;
            else { h = SW(c, a, 0); if (K::G) { q0 = SW(c, a, mbase); q1 = SW(c, a, mbase + 1); } }
            uint32_t nreg = (h >> 9) & 0xff, nmsg = HDR_NMSG(h);
            uint32_t idx = 0;
            while (idx < nmsg && ((K::G && idx == 0 ? q0 : (uint32_t)SW(c, a, mbase + 2 * idx)) & 0xff) != tag) idx++;
            if (idx < nmsg) {
                uint32_t m0, m1;
                if (K::G && idx == 0) { m0 = q0; m1 = q1; } else { m0 = SW(c, a, mbase + 2 * idx); m1 = SW(c, a, mbase + 2 * idx + 1); }
                nmsg--;
                if (!K::G || idx != nmsg) {                   // swap_remove (the last entry onto itself: nothing moves)
                    SW(c, a, mbase + 2 * idx) = SW(c, a, mbase + 2 * nmsg);
                    SW(c, a, mbase + 2 * idx + 1) = SW(c, a, mbase + 2 * nmsg + 1);
                }
                u0.w = m1;
                if (K::FR && P.uses_rpc && tag >= MADSIM_TAG_RPC_FIRST) { u0.w = m1 & 0xff; RQ_SET(1, m1 >> 8); }
                u0.y = (u0.y & 0x00ffffffu) | (((m0 >> 8) & 0xff) << 24);
                u0.x |= TF_INBOX;
                SW(c, a, 0) = HDR_SET_NMSG(h, nmsg);
            } else if (nreg >= P.mbox_regs) {
                OVF_SET(L, REGS_FULL);
            } else {
                const uint32_t reg = tag | (slot << 8) | (rxseq << 16) | ((gen & 0xff) << 24);
                if (may_have_twin(u0.x, gen)) for (uint32_t i = 0; i < nreg; i++) if (SW(c, a, 2 + i) == reg) OVF_SET(L, OVF_MODEL);   // rxseq wrapped onto a dead twin
                SW(c, a, 2 + nreg) = reg;
                SW(c, a, 0) = (h & ~(0xffu << 9)) | ((nreg + 1) << 9);
            }
            sub = 1;
            first_poll = true;
            if (recv_timeout_poll()) { sub = 0; pc++; }
            first_poll = false;
        } else if (K::FR && op == MS_OP_RPC_CALL) {          // first poll of timeout(d, ep.call(dst, req)) / ep.call(dst, req)
            if (imm >> 8) {                                    // timeout()'s Sleep exists before the call is polled
                uint64_t d2 = sleep_deadline(L, L.clock + (uint64_t)(imm >> 8) * NS_PER_MS);
                if (K::G) { buf_store64(c.gs, gs_addr_task(c, slot, 2 * 16u + 8u), make_uint2((uint32_t)d2, (uint32_t)(d2 >> 32))); pp.d2lo = (uint32_t)d2; pp.d2hi = (uint32_t)(d2 >> 32); }
                else { uint4 u2 = TU(c, slot, 2); u2.z = (uint32_t)d2; u2.w = (uint32_t)(d2 >> 32); TU(c, slot, 2) = u2; }
            }
            (void)rng_next(L); rng_log<K>(c, L);               // rsp_tag = random::<u64>(): one with() (rand.rs:146-148)
            uint64_t d1 = rand_delay_deadline<K>(c, L);        // send_to_raw -> NetSim::send: rand_delay first
            u1.z = (uint32_t)d1; u1.w = (uint32_t)(d1 >> 32); u1_dirty = true;
            sub = 1;
            first_poll = true;
            if (rpc_call_poll()) { sub = 0; pc++; }            // (never on the first poll: 1 ms floor)
            first_poll = false;
        } else if (op == MS_OP_SLEEP_RAND) {        // sleep(thread_rng().gen_range(lo..hi)): [DEP A.3],
            const uint64_t* dp = P.dur_table + 4 * a;          // host-precomputed UniformDuration {mode, low, range, zone}
            uint64_t mode = dp[0], low = dp[1], range = dp[2], zone = dp[3], d;
            for (;;) {                                         // on the GlobalRng itself: one with() per attempt
                uint64_t v = rng_next(L);
                rng_log<K>(c, L);
                if (mode == 0) {
                    uint64_t m = (uint64_t)(uint32_t)(v >> 32) * (uint64_t)(uint32_t)range;
                    if ((uint32_t)m <= (uint32_t)zone) { d = low + (m >> 32); break; }
                } else if (v * range <= zone) { d = low + __umul64hi(v, range); break; }
            }
            deadline = sleep_deadline(L, L.clock + d);
            want_sleep = true;
        } else if (op == MS_OP_SLEEP || op == MS_OP_SLEEP_UNTIL) {
            uint64_t base = L.clock;
            if (K::FT && op == MS_OP_SLEEP_UNTIL) { uint4 u2 = TU(c, slot, 2); base = u64of(u2.x, u2.y); }
            deadline = sleep_deadline(L, base + (uint64_t)b * NS_PER_S + imm);
            want_sleep = true;
        } else {
            // ---- everything else: rare, control-plane ops ----
            bool done_guard = true;                    // MS_OP_DONE: an init task's guard drops before its exit() (below)
            switch (op) {
            case MS_OP_DONE:
                u0.y = pc | (sub << 16) | (from << 24);
                TU(c, slot, 0) = u0;
                if (u1_dirty) { tu1_store<K>(c, slot, u1); u1_dirty = false; }
                // an init task is `async move { future.await; h.exit() }` (runtime/mod.rs:362-370): the body's locals are gone
                // when `future.await` returns, on a node that is not killed yet; then Spawner::exit = NodeInfo::kill on the
                // info it was spawned with (task/mod.rs:657-661)
                if (K::FN && ((PROGW(c, u0.x >> 24) >> 8) & MADSIM_PROG_INIT) && (u1.y >> 24) == NODE_INFO_GEN(node)) {
                    task_drop_locals<K>(c, L, slot, u0.x);
                    task_drop_guard<K>(c, L, slot, u0.x >> 24);
                    NODEW(0) |= 1u << node;
                    if ((u1.y >> 24) == 0) NODEW(2) |= 1u << node;
                    info_kill<K>(c, L, node, u1.y >> 24);
                    done_guard = false;
                }
                if (K::G) u0.x = task_finish<K>(c, L, slot, H_COMPLETED, done_guard, FinishKnown{done_guard, u0.x, u1.x, (K::FC && P.uses_chan) ? cu0_get() : 0u, SwitchPrefetch<K>::ON, g_hw});
                else { task_finish<K>(c, L, slot, H_COMPLETED, done_guard); u0.x = TWORD(c, slot, 0, 0); }
                st = ST_FINISHED;
                break;
            case MS_OP_SPAWN: {
                // another node's program: NodeHandle::spawn; this node's: task::spawn under this task's OWN NodeInfo (task/mod.rs:592-599)
                const bool via = (PROGW(c, a) & 0xff) != node;
                // (the request the child takes — rpc.rs:170 — goes into its units as spawn_task writes them; this task's request word is
                //  requested with the spawn's own reads.  The poll's copies of this task's flag word and unit 1 are current: k_lifecycle.h SpawnInit)
                uint32_t child;
                if constexpr (K::G) {
                    const bool mreq = K::FR && P.uses_rpc && (b & MADSIM_SPAWN_MOVE_REQUEST);
                    const uint32_t req = mreq ? RQ_GET(0) : 0u;
                    child = spawn_task<K>(c, L, a, true, via, via ? -1 : (int)slot, SpawnInit{!via, u0.x, u1.y, mreq, u0.w, from, req,
                                                                                                       SwitchPrefetch<K>::ON && g_slot != ~0u, g_slot, g_old, g_gen, g_seq, g_killed});
                } else {
                    child = spawn_task<K>(c, L, a, true, via, via ? -1 : (int)slot);
                    if (K::FR && P.uses_rpc && (b & MADSIM_SPAWN_MOVE_REQUEST) && child != 0xffffffffu) {   // rpc.rs:170
                        TWORD(c, child, 0, 3) = u0.w;
                        TWORD(c, child, 0, 1) = (TWORD(c, child, 0, 1) & 0x00ffffffu) | (from << 24);
                        TWORD(c, child, P.rpc_unit, 0) = TWORD(c, slot, P.rpc_unit, 0);
                    }
                }
                if (K::FC && P.uses_chan && (b & 2) && child != 0xffffffffu) {   // `async move`: the (tx, rx) pair moves
                    uint32_t cx = cu0_get();
                    TWORD(c, child, c.P.chan_unit, 0) = cx & 0x1ff;
                    cu0_set(cx | 0xff);
                }
            }
                pc++;
                break;
            case MS_OP_BUILD:
                for (uint32_t p = 1; p < P.n_progs; p++) {
                    uint32_t pw = PROGW(c, p);
                    if ((pw & 0xff) == a && ((pw >> 8) & MADSIM_PROG_INIT) && !((pw >> 8) & MADSIM_PROG_PRE)) spawn_task<K>(c, L, p, false);
                }
                pc++;
                break;
            case MS_OP_JOIN: {                             // task/join.rs:59-72 + async-task poll_task
                // `handle.await` moves the JoinHandle into the await: the task it names NOW is the one awaited, whatever a later
                // spawn stores in handle[prog]; task_finish hands its outcome to this task.  The await's state lives in `sub`
                // with bit 7 set (SUB_JOIN_*), which keeps it out of stage [A]: that chain is hot, this op is rare.
                uint32_t hs;
                if (sub == SUB_JOIN_WAIT) { st = ST_PENDING; break; }        // woken for another reason (a stale timer)
                if (sub != 0) { hs = sub == SUB_JOIN_CANCELLED ? H_CANCELLED : H_COMPLETED; sub = 0; }
                else {
                    const uint32_t h = HW(a);
                    hs = h & 3;
                    if (hs == H_RUNNING) {
                        const uint32_t cs = (h >> 8) & 0xff;
                        const uint32_t link = TWORD(c, cs, 1, 0);
                        TWORD(c, cs, 1, 0) = (link & 0xff) | (slot << 8) | (gen << 16);   // register awaiter
                        sub = SUB_JOIN_WAIT;
                        st = ST_PENDING;
                        break;
                    }
                }
                if (hs == H_NONE || ((hs == H_CANCELLED) != ((b & 1) != 0))) st = ST_PANIC;     // .unwrap() / .unwrap_err()
                else pc++;
                break;
            }
            case MS_OP_YIELD:                              // [DEP tokio yield_now outside a runtime]
                sub = 1;
                u0.x |= TF_SCHED;                          // wake_by_ref while RUNNING
                st = ST_PENDING;
                break;
            case MS_OP_IPVS: {                              // IpVirtualServer::{add,del}_{service,server} (net/ipvs.rs:50-85)
                if (!K::FA) { st = ST_PANIC; break; }
                const uint32_t k = b & 7u;
                uint32_t d0 = IPVSW(2 * k), d1 = IPVSW(2 * k + 1), n = (d1 >> 16) & 0xf;
                bool ok = true;
                if (a == MADSIM_IPVS_ADD_SERVICE) { d0 = 0; d1 = 1u << 24; }                 // insert(addr, Service { servers: [], rr_index: 0 })
                else if (a == MADSIM_IPVS_DEL_SERVICE) d1 &= ~(1u << 24);                    // remove
                else if (!((d1 >> 24) & 1u)) ok = false;                                     // .expect("service not found")
                else if (a == MADSIM_IPVS_ADD_SERVER) {                                      // servers.push
                    if (n >= 6) OVF_SET(L, OVF_MODEL);                                       // (a seventh server: the seed's two state words are full — the oracle says MADSIM_UNSUPPORTED at the same call)
                    else {
                        if (n < 4) d0 |= (imm & 0xff) << (8 * n); else d1 |= (imm & 0xff) << (8 * (n - 4));
                        d1 += 1u << 16;
                    }
                } else {                                                                     // servers.retain(|addr| addr != server_addr)
                    const uint32_t gone = SOCKW(c, imm & 0xff);
                    uint32_t e0 = 0, e1 = d1 & 0xfff00000u, m = 0;
                    for (uint32_t j = 0; j < n; j++) {
                        const uint32_t sj = j < 4 ? (d0 >> (8 * j)) & 0xff : (d1 >> (8 * (j - 4))) & 0xff;
                        if (!addr_eq(SOCKW(c, sj), gone)) {
                            if (m < 4) e0 |= sj << (8 * m); else e1 |= sj << (8 * (m - 4));
                            m++;
                        }
                    }
                    d0 = e0; d1 = e1 | (m << 16);
                }
                if (!ok) { st = ST_PANIC; break; }
                IPVSW(2 * k) = d0; IPVSW(2 * k + 1) = d1;
                pc++;
                break;
            }
            case MS_OP_HOOK_REQ:                            // NetSim::hook_rpc_req (net/mod.rs:240-262): HashMap::insert
                if (!K::FR) { st = ST_PANIC; break; }
                HOOKW(a) = ((uint32_t)HOOKW(a) & ~0x3ffffu) | 1u | ((b & 1) << 1) | ((imm & 0xff) << 2) | ((b >> 8) << 10);
                pc++;
                break;
            case MS_OP_HOOK_RSP:                            // NetSim::hook_rpc_rsp (net/mod.rs:264-284)
                if (!K::FR) { st = ST_PANIC; break; }
                HOOKW(a) = ((uint32_t)HOOKW(a) & 0x3ffffu) | (1u << 18) | ((b & 1) << 19) | ((imm & 0xff) << 20);
                pc++;
                break;
            case MS_OP_PANIC:                               // its message code: what restart_on_panic_matching compares
                if (K::FN) {
                    uint32_t code = imm & 0xff;
                    if (a & 1) {       // panic!("{}", flag + imm): the message is the decimal text of the value, and the nodes' rows were
                        code = (uint32_t)GREGW(b & 3) + imm;     // evaluated for values up to panic_dyn_max only
                        if (code > P.panic_dyn_max) { OVF_SET(L, OVF_MODEL); code = MADSIM_PANIC_CODE_OTHER; }   // (the workload's own declaration: MADSIM_UNSUPPORTED on both sides)
                    }
                    L.panic_code = code;
                }
                st = ST_PANIC;
                break;
            case MS_OP_ABORT: {                            // AbortHandle::abort (task/join.rs:158-163)
                if (!K::FN) { st = ST_PANIC; break; }
                uint32_t h = HW(a);
                if ((h & 3) == H_RUNNING) {
                    uint32_t cs = (h >> 8) & 0xff;
                    if (cs == slot) u0.x |= TF_CANCEL | TF_SCHED;   // its OWN handle: this task's unit0 lives in registers during the
                    else {                                          // poll; woken while RUNNING = re-queued after it (and dropped)
                        TWORD(c, cs, 0, 0) |= TF_CANCEL;
                        wake<K>(c, L, cs, h >> 16);
                    }
                }
                pc++;
                break;
            }
            case MS_OP_KILL: case MS_OP_RESTART:
                if (!K::FN) { st = ST_PANIC; break; }
                u0.y = pc | (sub << 16) | (from << 24);     // this task may be woken/killed by the call: sync LDS first
                TU(c, slot, 0) = u0;
                if (op == MS_OP_KILL) node_kill<K>(c, L, a); else node_restart<K>(c, L, a);
                u0.x = TWORD(c, slot, 0, 0);
                pc++;
                break;
            case MS_OP_PAUSE:                               // task/mod.rs:404-410
                if (!K::FN) { st = ST_PANIC; break; }
                NODEW(1) |= 1u << a;
                pc++;
                break;
            case MS_OP_RESUME: {                            // task/mod.rs:413-424: parked Runnables go back, in order
                if (!K::FN) { st = ST_PANIC; break; }
                NODEW(1) &= ~(1u << a);
                if (P.uses_pause) {
                    uint32_t n = PAUSEW(0), keep = 0;
                    for (uint32_t i = 0; i < n; i++) {
                        uint32_t ps = PAUSEW(1 + i);
                        if ((PROGW(c, TWORD(c, ps, 0, 0) >> 24) & 0xff) == a) ready_push<K>(c, L, ps);
                        else { PAUSEW(1 + keep) = ps; keep++; }
                    }
                    PAUSEW(0) = keep;
                }
                pc++;
                break;
            }
            case MS_OP_ASSERT_EXIT:                         // Handle::is_exit (task/mod.rs:444-449)
                if (!K::FN) { st = ST_PANIC; break; }
                if (((NODEW(0) >> a) & 1) != (b & 1)) st = ST_PANIC; else pc++;
                break;
            case MS_OP_CSEND: {                            // PayloadSender::send (net/mod.rs:417-421)
                if (!K::FC) { st = ST_PANIC; break; }
                uint32_t cx = cu0_get();
                if ((cx & 0xff) == 0xff) { u0.w = MADSIM_VAL_RESET; pc++; break; }
                uint32_t id = cx & 0xff, side = (cx >> 8) & 1;
                uint32_t cw = CONNW(id, 0);
                const uint32_t r_pre = Hoist<K>::CHAN ? (uint32_t)CONNW(id, 1 + side) : 0;   // (global-state builds: the parked receiver's word with the header, not after the link test)
                // (... and that receiver's flag word, which the wake-up below wants, with the link test's destination header: the test draws and
                //  stores nothing a task's flags depend on; the receiver is parked, not this task)
                uint32_t wf_pre = 0;
                if (Hoist<K>::CHAN && (r_pre & 1)) wf_pre = TWORD(c, (r_pre >> 1) & 0xff, 0, 0);
                uint64_t arrive = chan_test_link<K>(c, L, cw, side);          // draws happen before the closed check
                if (arrive == CHAN_LINK_PANIC) { st = ST_PANIC; break; }
                if (!(cw & (1u << (14 + 2 * side)))) { u0.w = MADSIM_VAL_RESET; pc++; break; }   // ConnectionReset
                uint32_t qn = (cw >> (17 + 4 * side)) & 0xf;
                // (below the ceiling of 15 queued payloads a larger chan_queue lifts it; AT the ceiling the 16th leaves the model: the oracle's queue
                //  holds the same payloads, it says MADSIM_UNSUPPORTED at the same send)
                if (qn >= P.chan_queue) { OVF_SET(L, P.chan_queue >= MADSIM_MAX_CHAN_QUEUE ? OVF_MODEL : OVF_CAP); pc++; break; }
                uint32_t e = 3 + (side * P.chan_queue + qn) * 3;
                CONNW(id, e) = imm; CONNW(id, e + 1) = (uint32_t)arrive; CONNW(id, e + 2) = (uint32_t)(arrive >> 32);
                CONNW(id, 0) = (cw & ~(0xfu << (17 + 4 * side))) | ((qn + 1) << (17 + 4 * side));
                uint32_t r = Hoist<K>::CHAN ? r_pre : (uint32_t)CONNW(id, 1 + side);
                if (r & 1) {                                                        // mpsc wakes the parked receiver
                    CONNW(id, 1 + side) = 0;
                    if (Hoist<K>::CHAN) {
#ifdef MADSIM_EMU
                        if (wf_pre != (uint32_t)TWORD(c, (r >> 1) & 0xff, 0, 0)) OVF_SET(L, OVF_BUG);
#endif
                        wake_with<K>(c, L, (r >> 1) & 0xff, r >> 9, wf_pre);
                    } else wake<K>(c, L, (r >> 1) & 0xff, r >> 9);
                }
                pc++;
                break;
            }
            case MS_OP_CRECV: {                            // rx.recv().await (net/mod.rs:386), sub == 0 here
                if (!K::FC) { st = ST_PANIC; break; }
                uint32_t cx = cu0_get();
                if ((cx & 0xff) == 0xff) { u0.w = MADSIM_VAL_RESET; pc++; break; }
                uint32_t id = cx & 0xff, dir = 1 - ((cx >> 8) & 1);
                uint32_t cw = CONNW(id, 0);
                // (global-state builds: the oldest queued payload's three words go out with the header — one round trip, not two)
                const uint32_t e0 = 3 + dir * P.chan_queue * 3;
                uint32_t h0 = 0, h1 = 0, h2 = 0;
                if (Hoist<K>::CHAN) { h0 = CONNW(id, e0); h1 = CONNW(id, e0 + 1); h2 = CONNW(id, e0 + 2); }
                uint32_t qn = (cw >> (17 + 4 * dir)) & 0xf;
                if (qn == 0) {
                    if (!(cw & (1u << (13 + 2 * dir)))) { u0.w = MADSIM_VAL_RESET; pc++; break; }   // all senders gone
                    CONNW(id, 1 + dir) = 1u | (slot << 1) | (gen << 9);
                    st = ST_PENDING;
                    break;
                }
                if (!Hoist<K>::CHAN) { h0 = CONNW(id, e0); h1 = CONNW(id, e0 + 1); h2 = CONNW(id, e0 + 2); }
                uint4 u3 = make_uint4((cx & 0x1ff) | (1u << 16), h0, h1, h2);   // backoff = 1 ms
                for (uint32_t i = 1; i < qn; i++) {            // VecDeque::pop_front (an entry's three loads first, then its three stores)
                    const uint32_t m0 = CONNW(id, e0 + i * 3), m1 = CONNW(id, e0 + i * 3 + 1), m2 = CONNW(id, e0 + i * 3 + 2);
                    CONNW(id, e0 + (i - 1) * 3) = m0; CONNW(id, e0 + (i - 1) * 3 + 1) = m1; CONNW(id, e0 + (i - 1) * 3 + 2) = m2;
                }
                CONNW(id, 0) = (cw & ~(0xfu << (17 + 4 * dir))) | ((qn - 1) << (17 + 4 * dir));
                cu_set(u3);
                crecv_arm(u3);
                break;
            }
            case MS_OP_CCLOSE: {
                if (!K::FC) { st = ST_PANIC; break; }
                uint32_t cx = cu0_get();
                if ((cx & 0xff) != 0xff) { conn_drop_handles<K>(c, L, cx & 0xff, (cx >> 8) & 1, (u0.x & TF_KILLED) != 0); cu0_set(cx | 0xff); }
                pc++;
                break;
            }
            case MS_OP_GSET: GREGW(a & 3) = imm; pc++; break;
            case MS_OP_GADD: if constexpr (K::G) GREGW(a & 3).add(imm); else GREGW(a & 3) += imm; pc++; break;      // (global-state builds: an add in memory, no round trip — k_mem.h buf_add32)
            case MS_OP_ASSERT_G: if (GREGW(a & 3) != imm) st = ST_PANIC; else pc++; break;
            case MS_OP_PANIC_IF_G_LT: if (GREGW(a & 3) < imm) st = ST_PANIC; else pc++; break;
            case MS_OP_MARK:
                if (!K::FT) { st = ST_PANIC; break; }     // t0 family and advance(): extended variant only
                TU(c, slot, 2) = make_uint4((uint32_t)L.clock, (uint32_t)(L.clock >> 32), 0, 0);
                pc++;
                break;
            case MS_OP_ASSERT_ELAPSED: {
                if (!K::FT) { st = ST_PANIC; break; }
                uint4 u2 = TU(c, slot, 2);
                uint64_t el = L.clock - u64of(u2.x, u2.y), d = (uint64_t)b * NS_PER_S + imm;
                bool ok = a == 0 ? el == d : a == 1 ? el >= d : el < d;
                if (!ok) st = ST_PANIC; else pc++;
                break;
            }
            case MS_OP_ADVANCE:                            // time/mod.rs:103-106
                if (!K::FT) { st = ST_PANIC; break; }
                if (K::G && L.pq_n) break;                 // pushes of this round are still queued: flush at the head, then come back here
                if (K::NH && (uint64_t)b * NS_PER_S + imm >= NH_HORIZON) OVF_SET(L, OVF_CAP);   // (the 8-byte entries' deadlines are read relative to the clock)
                L.clock += (uint64_t)b * NS_PER_S + imm;
                pc++;
                u0.y = pc | (sub << 16) | (from << 24);
                TU(c, slot, 0) = u0;
                if (u1_dirty) { tu1_store<K>(c, slot, u1); u1_dirty = false; }
                timer_expire<K>(c, L, L.clock);
                heap_pre.idx = ~0u;                          // (the heap has changed)
                u0 = TU(c, slot, 0); u1 = load_u1<K>(c, slot);
                if (HoistRpc<K>::ON && P.uses_rpc) { pp.rq0 = TWORD(c, slot, P.rpc_unit, 0); pp.rq1 = TWORD(c, slot, P.rpc_unit, 1); }   // (a delivery to this task stages its rsp_tag)
                from = u0.y >> 24;
                break;
            case MS_OP_CLOSE: {
                uint32_t h = SW(c, a, 0);
                if (K::LIFE) {
                    if (sock_owned_by<K>(c, a, h, slot, gen)) endpoint_drop<K>(c, L, a, (u0.x & TF_KILLED) != 0);
                } else if ((h & 1) && sock_owned_by<K>(c, a, h, slot, gen)) SW(c, a, 0) = h & ~1u;
                pc++;
                break;
            }
            case MS_OP_CLOG_NODE:
                if (b & 1) CLOGW(0) |= 1u << a;
                if (b & 2) CLOGW(1) |= 1u << a;
                if (K::LIFE && MADSIM_CLOG_MIRROR) L.loss_always |= 0x100u;          // "a node is clogged" mirrored in the lane (k_channel.h net_try_send)
                pc++;
                break;
            case MS_OP_UNCLOG_NODE:
                if (b & 1) CLOGW(0) &= ~(1u << a);
                if (b & 2) CLOGW(1) &= ~(1u << a);
                if (K::LIFE && MADSIM_CLOG_MIRROR && ((uint32_t)CLOGW(0) | (uint32_t)CLOGW(1)) == 0) L.loss_always &= ~0x100u;
                pc++;
                break;
            case MS_OP_CLOG_LINK:
                CLOGW(2 + a) |= 1u << b;
                if (K::LIFE && MADSIM_CLOG_MIRROR) L.loss_always |= 0x200u;          // "a link has been clogged": stays set (unclog_link does not look at the other rows)
                pc++;
                break;
            case MS_OP_UNCLOG_LINK:
                CLOGW(2 + a) &= ~(1u << b);
                pc++;
                break;
            case MS_OP_RANDOM: {                            // one with() on the GlobalRng's RngCore impl (rand.rs:142-158)
                uint64_t v = rng_next(L); rng_log<K>(c, L);
                u0.w = a == 0 ? (uint32_t)(v >> 32) : (uint32_t)((v >> 32) & 0xff);   // gen::<u32>() / getrandom 1 byte [DEP]
                pc++;
                break;
            }
            case MS_OP_TRACE_TIME: {                        // SystemTime::now() / Instant elapsed (time/system_time.rs)
                if (!K::LIFE) { st = ST_PANIC; break; }
                uint64_t v = a == 2 ? (uint64_t)u0.w : L.clock;
                if (a == 0) v += (1639872000ull + NODEW(4 + ((P.n_nodes + 4) >> 2))) * NS_PER_S;   // 52 x 365 days + the draw
                L.obs_hash = (L.obs_hash ^ v) * FNV_PRIME;
                pc++;
                break;
            }
            case MS_OP_RAND_BOOL:                           // thread_rng().gen_bool(p) [DEP A.4]
                u0.w = gen_bool_pint<K>(c, L, loss_pint_at(P, a), loss_always_at(P, a)) ? 1u : 0u;
                pc++;
                break;
            case MS_OP_SET_LOSS:
                L.loss_pint = loss_pint_at(P, a);
                L.loss_always = K::LIFE ? ((L.loss_always & ~1u) | loss_always_at(P, a)) : loss_always_at(P, a);
                pc++;
                break;
            default:
                // NetSim::update_config(|c| c.send_latency = ..) (net/mod.rs:138-141, network.rs:129): extended builds only — geometry.h routes
                // every workload with the op to one.  (Under `default`, not a case of its own: the base-op builds' switch — the headline
                // kernel's — keeps round 5's code byte for byte; a case label here cost it 1.4 % on the GPU, profiles/r6_experiments.md.)
                if (K::LIFE && op == MS_OP_SET_LATENCY) { L.loss_always = (L.loss_always & ~0x70u) | (((a & 3u) + 1u) << 4); pc++; }
                else st = ST_PANIC;
                break;
            }
        }
        PROBE(8);
        if (want_delay) {                                  // NetSim::rand_delay (net/mod.rs:287-292)
            REG(15);
            deadline = rand_delay_deadline<K>(c, L);
            want_sleep = true;
        }
        if (want_sleep) {                                  // first Sleep::poll: never elapsed (1 ms floor)
            REG(17);
            u1.z = (uint32_t)deadline; u1.w = (uint32_t)(deadline >> 32); u1_dirty = true;
            sub = (op == MS_OP_RECV) ? 3 : 1;
            timer_schedule<K>(c, L, deadline, (EV_WAKE << EV_SHIFT) | (gen << 8) | slot, 0, true);
            st = ST_PENDING;
        }
    }
    PROBE(9);
    if (st != ST_FINISHED) {
        u0.y = pc | (sub << 16) | (from << 24);
        if (u1_dirty) tu1_store<K>(c, slot, u1);
    }
    return st == ST_PANIC;
}

}  // namespace madsim_k

#endif
