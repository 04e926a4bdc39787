// k_rng.h — GlobalRng: xoshiro256++, rand 0.8 gen_range / gen_bool / UniformDuration, determinism log.
// Part of sim_kernel.hip (included in this order: k_state, k_rng, k_timer, k_net, k_lifecycle, k_channel, k_poll).
#ifndef MADSIM_K_RNG_H
#define MADSIM_K_RNG_H

namespace madsim_k {

// ---- GlobalRng ---------------------------------------------------------------------------------
// Xoshiro256PlusPlus::next_u64 [DEP rand_xoshiro 0.6] without the call counter (rng_next below counts; the rejection loops
// count their trips in a 32-bit register — the compiler keeps it scalar, a copy per trip — and add once per draw.  Round 6 measured a per-lane 32-bit
// counter inside the loops, folded into the 64-bit field every 16th pass, instead: +0.8 % time on the headline kernel, profiles/r6_ab_rng_count_32bit.txt)
__device__ __forceinline__ uint64_t rng_out(const Lane& L) { return add64_1(rotl64<23>(L.s0 + L.s3), L.s0); }
__device__ __forceinline__ void rng_advance(Lane& L) {
    // s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= s1 << 17 with the two three-input xors as one v_bitop3_b32 per half
    const uint64_t t = shl64<17>(L.s1);
    const uint64_t d1 = L.s3 ^ L.s1;
    const uint64_t b1 = xor3_64(L.s1, L.s2, L.s0);
    const uint64_t c2 = xor3_64(L.s2, L.s0, t);
    L.s0 ^= d1; L.s1 = b1; L.s2 = c2;
    L.s3 = rotl64<45>(d1);
}
__device__ __forceinline__ uint64_t rng_step(Lane& L) {
    const uint64_t r = rng_out(L);
    rng_advance(L);
    return r;
}
__device__ __forceinline__ uint64_t rng_next(Lane& L) {
    L.rng_calls++;
    return rng_step(L);
}
// Builds that keep the determinism log hold the output of the CURRENT state in Lane::peek: the log's entry for a with() is a byte of
// `rng.clone().gen()` (rand.rs:64-88), i.e. of the next with()'s first output, so rng_log computes that number anyway.  The draws whose accepted output is not wanted
// behind their loop start from it (gen_index at queue length 1, gen_range_small for general ranges, gen_bool, the latency draw): level 2; level 1 = gen_index only.
template <class K> struct Peek { static constexpr bool ON = MADSIM_RNG_PEEK && MADSIM_K_LOG_ENABLED && !K::NOLOG && (!K::LIFE || MADSIM_RNG_PEEK_LIFE), ALL = ON && MADSIM_RNG_PEEK > 1; };

// One with()'s rejection loop: outputs until `rej` accepts one.
// (Round 6 measured the first trip PEELED — it takes the peek and only advances, four VALU fewer per with() — on the headline kernel: +4 % time,
//  timer storm +1.7 %.  The loop entered under the rejected lanes' mask pays more in phi copies of the generator state than the peel saves;
//  profiles/r6_ab_rng_peek.txt.  The form that keeps the loop's shape is gen_index's below.)
template <class K, int REGION, class Rej>
__device__ __forceinline__ uint64_t rng_draw(Lane& L, uint32_t& trips, Rej rej) {
    uint64_t v;
    trips = 0;
    do { REG(REGION); v = rng_step(L); trips++; } while (rej(v));
    return v;
}

// rand 0.8's accept test `lo64(v * range) <= zone` for a range below 2^32: zone = (range << lz) - 1 has its low word all
// ones, so only the high word of the low 64 product bits is compared: (v.hi * range + mulhi(v.lo, range)) mod 2^32.
__device__ __forceinline__ bool reject32(uint64_t v, uint32_t range, uint32_t zone_hi) {
    return (uint32_t)(v >> 32) * range + __umulhi((uint32_t)v, range) > zone_hi;
}
// ... and for a compile-time range of the form 2^k + 1 (k <= 4) the low 64 product bits are ONE v_lshl_add_u64, (v << k) + v,
// instead of a v_mul_hi_u32 and a v_mad_u64_u32: NetSim::rand_delay's gen_range(0..5) runs ~5 trips per executor iteration
template <uint32_t RANGE>
__device__ __forceinline__ bool reject_const(uint64_t v, uint32_t zone_hi) {
    if (RANGE == 3 || RANGE == 5 || RANGE == 9 || RANGE == 17)
        return (uint32_t)(mul_pow2p1<RANGE == 3 ? 1 : RANGE == 5 ? 2 : RANGE == 9 ? 3 : 4>(v) >> 32) > zone_hi;
    return reject32(v, RANGE, zone_hi);
}

// One determinism-log byte per GlobalRng::with (rand.rs:64-88): clone.gen::<u8>() ^ xor-fold(elapsed).
template <class K>
__device__ __forceinline__ void rng_log(const Ctx& c, Lane& L, bool have = false) {     // have: Lane::peek is current (gen_index)
    if (!MADSIM_K_LOG_ENABLED || K::NOLOG) return;
    if (Peek<K>::ON && !have) L.peek = rng_out(L);                  // what the clone's next_u64 would return = the next with()'s first output
    if (K::LOGSW && c.P.no_log) return;                    // (wave-uniform)
    const uint64_t r = Peek<K>::ON ? L.peek : rng_out(L);
    uint32_t v = (uint32_t)(r >> 32);
    uint32_t f = (uint32_t)L.clock ^ (uint32_t)(L.clock >> 32);
    f ^= f >> 16; f ^= f >> 8;
    v = (v ^ f) & 0xff;
    L.trace_hash = (L.trace_hash ^ v) * FNV_PRIME;
    if (K::TRACE) { if (L.log_len < c.P.trace_cap) c.tlog[L.log_len] = (uint8_t)v; }
    L.log_len++;
}

// gen_range(lo..hi) on u64 [DEP rand 0.8 UniformInt::sample_single_inclusive]; one with() per call.
template <class K>
__device__ __forceinline__ uint64_t gen_range_u64(const Ctx& c, Lane& L, uint64_t lo, uint64_t range) {
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    uint64_t v;
    uint32_t trips;
    v = rng_draw<K, 16>(L, trips, [&](uint64_t x) { return x * range > zone; });
    L.rng_calls += trips;
    rng_log<K>(c, L);
    return lo + __umul64hi(v, range);
}

// ready-queue index draw: range = len <= 255, so the 128-bit product splits into two 32x32 pieces.
template <class K>
__device__ __forceinline__ uint32_t gen_index(const Ctx& c, Lane& L, uint32_t len) {
    uint64_t v;
    uint32_t trips = 0;
    // The queue usually holds ONE task: range 1, zone 2^63 - 1 — the draw is rejected while the output's top bit is set and the
    // index is 0.  When that is so for every lane of the wave (k_mem.h wave_all) the accept test is one signed compare instead of
    // the two multiplies of reject32: ~7 wave trips an executor pass end here.  (Same outputs consumed, same results.)
    if (wave_all(len == 1)) {
        if (Peek<K>::ON) {
            // the accepted output itself is not needed (the index is 0): each trip tests the output in hand, advances and computes the next one —
            // the do-while it always was, and the output left in hand at the exit is the log entry's number (rng_log_with)
            uint64_t p = L.peek;
#ifdef MADSIM_EMU
            if (p != rng_out(L)) OVF_SET(L, OVF_BUG);          // every with() of these builds ends in rng_log: the peek is the current state's output
#endif
            bool rej;
            do { REG(1); rej = EXP_ACCEPT((int32_t)(uint32_t)(p >> 32) < 0); rng_advance(L); p = rng_out(L); trips++; } while (rej);
            L.peek = p;
            L.rng_calls += trips;
            rng_log<K>(c, L, true);
            return 0;
        }
        v = rng_draw<K, 1>(L, trips, [&](uint64_t x) { return EXP_ACCEPT((int32_t)(uint32_t)(x >> 32) < 0); });
        L.rng_calls += trips;
        rng_log<K>(c, L);
        return 0;
    }
    const uint32_t zone_hi = (len << (__builtin_clz(len))) - 1;
    v = rng_draw<K, 1>(L, trips, [&](uint64_t x) { return EXP_ACCEPT(reject32(x, len, zone_hi)); });
    L.rng_calls += trips;
    rng_log<K>(c, L);
    uint64_t mid = (uint64_t)(uint32_t)(v >> 32) * len + (((uint64_t)(uint32_t)v * len) >> 32);
    return (uint32_t)(mid >> 32);                               // high 64 bits of v * len
}

// gen_range with a compile-time range < 2^32.
template <class K, uint32_t RANGE>
__device__ __forceinline__ uint32_t gen_range_small(const Ctx& c, Lane& L) {
    constexpr uint64_t zone = ((uint64_t)RANGE << __builtin_clzll((uint64_t)RANGE)) - 1;
    static_assert((uint32_t)zone == 0xffffffffu, "reject32 compares the high word only");
    uint64_t v;
    uint32_t trips = 0;
    constexpr bool POW2P1 = RANGE == 3 || RANGE == 5 || RANGE == 9 || RANGE == 17;
    if (Peek<K>::ALL && !POW2P1) {
        // gen_index's form: test the output in hand, advance, compute the next one.  The accept test's operand — the middle 64 bits of the 128-bit
        // product v * RANGE — holds the RESULT in its high word, so the accepted output itself is not needed behind the loop.
        // (ranges 2^k + 1 test with one shift-add and compute the result from the output afterwards: they keep the form below)
        uint64_t p = L.peek, mid;
        bool rej;
        do {
            REG(RANGE == 50 ? 18 : 16);
            mid = (uint64_t)(uint32_t)(p >> 32) * RANGE + __umulhi((uint32_t)p, RANGE);
            rej = EXP_ACCEPT((uint32_t)mid > (uint32_t)(zone >> 32));
            rng_advance(L); p = rng_out(L); trips++;
        } while (rej);
        L.peek = p;
        L.rng_calls += trips;
        rng_log<K>(c, L, true);
        return (uint32_t)(mid >> 32);
    }
    v = rng_draw<K, RANGE == 50 ? 18 : 16>(L, trips, [&](uint64_t x) { return EXP_ACCEPT(reject_const<RANGE>(x, (uint32_t)(zone >> 32))); });
    L.rng_calls += trips;
    rng_log<K>(c, L);
    uint64_t mid = (uint64_t)(uint32_t)(v >> 32) * RANGE + (((uint64_t)(uint32_t)v * RANGE) >> 32);
    return (uint32_t)(mid >> 32);
}

// gen_bool through GlobalRng's RngCore impl (rand.rs:142-158): one with() per draw [DEP Bernoulli].
template <class K>
__device__ __forceinline__ bool gen_bool_pint(const Ctx& c, Lane& L, uint64_t p_int, uint32_t always) {
    if (always) return true;
    REG(7);
    if (Peek<K>::ALL) {                                    // the output in hand, then the next one for the log entry (gen_index)
        const uint64_t v = L.peek;
        L.rng_calls++; rng_advance(L); L.peek = rng_out(L);
        rng_log<K>(c, L, true);
        return v < p_int;
    }
    uint64_t v = rng_next(L);
    rng_log<K>(c, L);
    return v < p_int;
}

// UniformDuration sample on the GlobalRng itself (network.rs:267): one with() per attempt [DEP A.3].
// The range is `self.config.send_latency` at the moment of the call: the launch's (KParams.lat_*), or — extended builds, workloads
// with MS_OP_SET_LATENCY — the lat_table entry the seed's last NetSim::update_config named (Lane::loss_always bits 4-6).
// (select chains, not P.lat_tab_low[k]: a per-lane index into the by-value kernel-argument block would copy the block to scratch)
template <class K>
__device__ __forceinline__ uint64_t sample_latency(const Ctx& c, Lane& L) {
    const KParams& P = c.P;
    uint32_t mode = P.lat_mode;
    uint64_t low = P.lat_low, range = P.lat_range, zone = P.lat_zone;
    if (K::LIFE && P.uses_set_lat) {
        const uint32_t k = (L.loss_always >> 4) & 7u;
        mode = k == 0 ? mode : k == 1 ? P.lat_tab_mode[0] : k == 2 ? P.lat_tab_mode[1] : k == 3 ? P.lat_tab_mode[2] : P.lat_tab_mode[3];
        low = k == 0 ? low : k == 1 ? P.lat_tab_low[0] : k == 2 ? P.lat_tab_low[1] : k == 3 ? P.lat_tab_low[2] : P.lat_tab_low[3];
        range = k == 0 ? range : k == 1 ? P.lat_tab_range[0] : k == 2 ? P.lat_tab_range[1] : k == 3 ? P.lat_tab_range[2] : P.lat_tab_range[3];
        zone = k == 0 ? zone : k == 1 ? P.lat_tab_zone[0] : k == 2 ? P.lat_tab_zone[1] : k == 3 ? P.lat_tab_zone[2] : P.lat_tab_zone[3];
    }
    uint64_t res;
    bool ok;
    do {                                                // (one exit: see k_main.h on exit edges)
        REG(8);
        uint64_t v;
        if (Peek<K>::ALL) { v = L.peek; L.rng_calls++; rng_advance(L); L.peek = rng_out(L); rng_log<K>(c, L, true); }
        else { v = rng_next(L); rng_log<K>(c, L); }
        if (mode == 0) {
            uint64_t m = (uint64_t)(uint32_t)(v >> 32) * (uint64_t)(uint32_t)range;
            ok = (uint32_t)m <= (uint32_t)zone;
            res = low + (m >> 32);
        } else {
            ok = v * range <= zone;
            res = low + __umul64hi(v, range);
        }
    } while (!ok);
    return res;
}

}  // namespace madsim_k

#endif
