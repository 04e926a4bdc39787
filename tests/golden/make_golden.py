#!/usr/bin/env python3
"""Generates tests/golden/*.json — known-answer vectors for the CPU oracle.

This is an INDEPENDENT second restatement (pure Python, arbitrary-precision ints, dict/list based,
written from SURVEY.md Appendix A and the reference sources, not from oracle/madsim_oracle.c) of
  - rand_xoshiro's Xoshiro256PlusPlus + SplitMix64 seeding, rand 0.8 gen_range / Bernoulli /
    UniformDuration                                                  [DEP, SURVEY A.1-A.4]
  - the executor loop of madsim/src/sim/task/mod.rs:220-323, time/mod.rs:45-124, time/sleep.rs:47-54,
    utils/mpsc.rs:73-83, net/mod.rs:287-333, net/network.rs:261-313, net/endpoint.rs:331-362
for the small op subset the fixtures use.  It cannot be pinned to the Rust reference in this image
(no rustc/cargo; the reference ships no golden vectors for this path), so what the fixtures give is
agreement between two independently written restatements plus the public xoshiro256++ / SplitMix64
known answers quoted in SURVEY.md Appendix B.

Run:  python tests/golden/make_golden.py     (rewrites the .json files next to it)
"""
import heapq  # noqa: F401  (deliberately NOT used: Rust's BinaryHeap order is restated by hand)
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from madsim_amd import _abi as A  # noqa: E402
from madsim_amd import workload as W  # noqa: E402

M64 = (1 << 64) - 1
FNV_OFFSET, FNV_PRIME = 14695981039346656037, 1099511628211
OPN = {v: k for k, v in A.OP.items()}


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


class Xoshiro:
    def __init__(self, seed=None, state=None):
        if state is not None:
            self.s = list(state)
        else:
            x, self.s = seed, []
            for _ in range(4):
                x = (x + 0x9E3779B97F4A7C15) & M64
                z = x
                z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
                z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
                self.s.append(z ^ (z >> 31))
        self.calls = 0

    def peek(self):
        s = self.s
        return (rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64

    def next(self):
        s = self.s
        r = self.peek()
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        self.calls += 1
        return r


def gen_range_attempts(rng, lo, hi):
    """gen_range(lo..hi) for u64: returns (value, attempts)."""
    rg = hi - lo
    lz = 64 - rg.bit_length()
    zone = ((rg << lz) - 1) & M64
    n = 0
    while True:
        v = rng.next(); n += 1
        m = v * rg
        if (m & M64) <= zone:
            return lo + (m >> 64), n


def duration_params(lo, hi):
    S = 10**9
    h = hi - 1
    lo_s, lo_n, hi_s, hi_n = lo // S, lo % S, h // S, h % S
    if hi_n < lo_n:
        hi_s -= 1; hi_n += S
    if lo_s == hi_s:
        r = hi_n - lo_n + 1
        reject = (2**32 - r) % r
        return 0, lo_s * S + lo_n, r, 2**32 - 1 - reject
    r = h - lo + 1
    reject = (2**64 - r) % r
    return 1, lo, r, 2**64 - 1 - reject


class Sim:
    """One seed.  Structures: ready list of task dicts, heap list of [deadline, cb], per-socket mailbox."""

    def __init__(self, w, cfg, seed):
        self.insns = [(w.insns[i].op, w.insns[i].a, w.insns[i].b, w.insns[i].imm) for i in range(w.struct.n_insns)]
        self.progs = [(w.progs[i].node, w.progs[i].flags, w.progs[i].entry) for i in range(w.struct.n_progs)]
        self.socks = [(w.socks[i].node, w.socks[i].port) for i in range(w.struct.n_socks)]
        self.cfg = cfg
        self.rng = Xoshiro(seed)
        self.clock = 0
        self.log = []
        self.logging = False
        self.heap = []
        self.ready = []
        self.tasks = []
        self.handles = {}
        self.bound = {}          # addr index -> mailbox dict
        self.steps = 0
        self.msg_count = 0
        self.obs = FNV_OFFSET
        self.p_loss = None if cfg.packet_loss_rate == 1.0 else int(cfg.packet_loss_rate * 2.0**64)
        self.lat = duration_params(cfg.lat_lo_ns, cfg.lat_hi_ns)

    # -- GlobalRng::with ---------------------------------------------------------------------
    def with_log(self):
        if not self.logging:
            return
        v = (self.rng.peek() >> 32) & 0xFF
        t = self.clock
        for i in range(8):
            v ^= (t >> (8 * i)) & 0xFF
        self.log.append(v)

    def gen_range(self, lo, hi):
        v, _ = gen_range_attempts(self.rng, lo, hi)
        self.with_log()
        return v

    def latency(self):
        mode, low, rg, zone = self.lat
        while True:
            v = self.rng.next(); self.with_log()
            if mode == 0:
                m = (v >> 32) * rg
                if (m & 0xFFFFFFFF) <= zone:
                    return low + (m >> 32)
            else:
                m = v * rg
                if (m & M64) <= zone:
                    return low + (m >> 64)

    # -- BinaryHeap with reversed deadline order -----------------------------------------------
    def sift_up(self, pos):
        h = self.heap
        hole = h[pos]
        while pos > 0:
            parent = (pos - 1) // 2
            if hole[0] >= h[parent][0]:
                break
            h[pos] = h[parent]; pos = parent
        h[pos] = hole

    def timer_add(self, deadline, cb):
        self.heap.append([deadline, cb]); self.sift_up(len(self.heap) - 1)

    def timer_pop(self):
        h = self.heap
        item = h.pop()
        if h:
            item, h[0] = h[0], item
            end, pos = len(h), 0
            hole = h[0]
            child = 1
            while child + 1 < end:
                if h[child][0] >= h[child + 1][0]:
                    child += 1
                h[pos] = h[child]; pos = child; child = 2 * pos + 1
            if child == end - 1:
                h[pos] = h[child]; pos = child
            h[pos] = hole
            self.sift_up(pos)
        return item

    def expire(self, now):
        while self.heap and self.heap[0][0] <= now:
            _, cb = self.timer_pop()
            self.steps += 1
            cb()

    # -- tasks -----------------------------------------------------------------------------------
    def spawn(self, prog):
        t = dict(prog=prog, node=self.progs[prog][0], pc=self.progs[prog][2], sub=0, alive=True, sched=True,
                 running=False, joiner=None, cnt=[0, 0], val=0, frm=0, inbox=None, deadline=0, t0=0, owned=[])
        self.tasks.append(t); self.ready.append(t); self.handles[prog] = t
        return t

    def wake(self, t):
        if not t["alive"] or t["sched"]:
            return
        t["sched"] = True
        if not t["running"]:
            self.ready.append(t)

    def finish(self, t):
        for a in t["owned"]:
            if self.bound.get(a) is not None and self.bound[a]["owner"] is t:
                self.bound[a] = None
        t["alive"] = False
        if t["joiner"] is not None:
            self.wake(t["joiner"])

    def sleep_deadline(self, d):
        return max(d, self.clock + 1_000_000)

    def rand_delay(self):
        delay = self.gen_range(0, 5) * 1000
        return self.sleep_deadline(self.clock + delay)

    def sleep_poll(self, t):
        if self.clock >= t["deadline"]:
            return True
        self.timer_add(t["deadline"], lambda t=t: self.wake(t))
        return False

    def deliver(self, mbox, tag, val, frm):
        i = 0
        regs = mbox["regs"]
        while i < len(regs):
            if regs[i][0] == tag:
                _, rt = regs[i]
                regs[i] = regs[-1]; regs.pop()
                if rt["alive"] and rt["inbox"] is None:
                    rt["inbox"] = (val, frm); self.wake(rt)
                    return
            else:
                i += 1
        mbox["msgs"].append((tag, val, frm))

    def poll(self, t):
        """Returns 'panic' or None."""
        while True:
            op, a, b, imm = self.insns[t["pc"]]
            name = OPN[op]
            dur = b * 10**9 + imm
            if name == "DONE":
                self.finish(t); return None
            elif name == "SPAWN":
                self.spawn(a); t["pc"] += 1
            elif name == "JOIN":
                c = self.handles[a]
                if c["alive"]:
                    c["joiner"] = t; return None
                t["pc"] += 1
            elif name == "YIELD":
                if t["sub"] == 0:
                    t["sub"] = 1; self.wake(t); return None
                t["sub"] = 0; t["pc"] += 1
            elif name == "SET":
                t["cnt"][a & 1] = imm & 0xFFFF; t["pc"] += 1
            elif name == "DJNZ":
                t["cnt"][a & 1] = (t["cnt"][a & 1] - 1) & 0xFFFF
                t["pc"] = b if t["cnt"][a & 1] else t["pc"] + 1
            elif name == "TRACE":
                v = imm + (t["cnt"][a & 1] if b & 1 else 0)
                self.obs = ((self.obs ^ v) * FNV_PRIME) & M64; t["pc"] += 1
            elif name == "SLEEP":
                if t["sub"] == 0:
                    t["deadline"] = self.sleep_deadline(self.clock + dur); t["sub"] = 1
                if not self.sleep_poll(t):
                    return None
                t["sub"] = 0; t["pc"] += 1
            elif name == "MARK":
                t["t0"] = self.clock; t["pc"] += 1
            elif name == "ASSERT_ELAPSED":
                el = self.clock - t["t0"]
                if not {0: el == dur, 1: el >= dur, 2: el < dur}[a]:
                    return "panic"
                t["pc"] += 1
            elif name in ("BIND", "SEND", "REPLY"):
                if t["sub"] == 0:
                    t["deadline"] = self.rand_delay(); t["sub"] = 1
                if not self.sleep_poll(t):
                    return None
                if name == "BIND":
                    if self.socks[a][0] != t["node"] or self.bound.get(a) is not None:
                        return "panic"
                    self.bound[a] = dict(owner=t, regs=[], msgs=[]); t["owned"].append(a)
                else:
                    dst = (b & 0xFF) if name == "SEND" else t["frm"]
                    lost = True if self.p_loss is None else None
                    if lost is None:
                        v = self.rng.next(); self.with_log()
                        lost = v < self.p_loss
                    if not lost:
                        self.msg_count += 1
                        lat = self.latency()
                        mbox = self.bound.get(dst)
                        if mbox is not None:
                            self.timer_add(self.clock + lat,
                                           lambda m=mbox, tag=b >> 8, val=imm, frm=a, d=dst:
                                           self.deliver(m, tag, val, frm) if self.bound.get(d) is m else None)
                t["sub"] = 0; t["pc"] += 1
            elif name == "RECV":
                mbox = self.bound[a]
                tag = b >> 8
                if t["sub"] == 0:
                    t["inbox"] = None
                    idx = next((i for i, m in enumerate(mbox["msgs"]) if m[0] == tag), None)
                    if idx is not None:
                        m = mbox["msgs"][idx]
                        mbox["msgs"][idx] = mbox["msgs"][-1]; mbox["msgs"].pop()
                        t["inbox"] = (m[1], m[2])
                    else:
                        mbox["regs"].append((tag, t))
                    t["sub"] = 1
                if t["sub"] == 1:
                    if t["inbox"] is None:
                        return None
                    t["val"], t["frm"] = t["inbox"]; t["inbox"] = None
                    t["deadline"] = self.rand_delay(); t["sub"] = 2
                if not self.sleep_poll(t):
                    return None
                t["sub"] = 0; t["pc"] += 1
            elif name == "ASSERT_VAL":
                if t["val"] != imm:
                    return "panic"
                t["pc"] += 1
            elif name == "PANIC":
                return "panic"
            else:
                raise NotImplementedError(name)

    def run(self, time_limit=0):
        self.gen_range(0, 60 * 60 * 24 * 365)       # TimeRuntime::new base_time (not logged)
        self.logging = True
        main = self.spawn(0)
        verdict = A.PASS
        while True:
            panicked = False
            while self.ready:
                idx = self.gen_range(0, len(self.ready))
                t = self.ready[idx]
                self.ready[idx] = self.ready[-1]; self.ready.pop()
                self.steps += 1
                t["sched"] = False; t["running"] = True
                if self.poll(t) == "panic":
                    panicked = True; break
                if t["alive"]:
                    t["running"] = False
                    if t["sched"]:
                        self.ready.append(t)
                self.clock += self.gen_range(50, 100)
                self.expire(self.clock)
            if panicked:
                verdict = A.PANIC; break
            if not main["alive"]:
                break
            if not self.heap:
                verdict = A.DEADLOCK; break
            t = self.heap[0][0] + 50
            self.expire(t)
            self.clock = t
            if time_limit and self.clock >= time_limit:
                verdict = A.TIME_LIMIT; break
        h = FNV_OFFSET
        for v in self.log:
            h = ((h ^ v) * FNV_PRIME) & M64
        return dict(verdict=verdict, steps=self.steps, clock_ns=self.clock, msg_count=self.msg_count,
                    rng_calls=self.rng.calls, trace_hash=h, obs_hash=self.obs, log=bytes(self.log).hex())


def workloads():
    out = {}
    wl = W.WorkloadBuilder(); m = wl.main(); m.mark(); m.sleep(secs=1); m.assert_elapsed("==", secs=1, ns=50); m.done()
    out["sleep_1s"] = wl.build()
    out["pingpong_2x3"] = W.pingpong(2, 3)
    out["pingpong_4x2"] = W.pingpong(4, 2)
    wl = W.WorkloadBuilder()
    ts = []
    for i in range(3):
        t = wl.task(0); t.set(0, 5); top = t.label(); t.trace(i * 10, add_reg=0); t.yield_now(); t.djnz(0, top); t.done(); ts.append(t)
    m = wl.main()
    for t in ts:
        m.spawn(t)
    for t in ts:
        m.join(t)
    m.done()
    out["yield_3x5"] = wl.build()
    # out-of-order tags (net/endpoint.rs send_recv test shape, without the Barrier)
    wl = W.WorkloadBuilder(); n1, n2 = wl.create_node(), wl.create_node(); a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
    s = wl.task(n1); s.bind(a1); s.sleep(ms=100); s.send_to(a1, a2, 1, 11); s.sleep(secs=1); s.send_to(a1, a2, 2, 22); s.done()
    r = wl.task(n2); r.bind(a2); r.recv_from(a2, 2); r.assert_val(22); r.recv_from(a2, 1); r.assert_val(11); r.done()
    m = wl.main(); m.spawn(s); m.spawn(r); m.join(r); m.done()
    out["tags_out_of_order"] = wl.build()
    return out


def main():
    kat = {"xoshiro_state_1234": [], "seed_from_u64": {}, "gen_range": [], "duration_params": []}
    x = Xoshiro(state=[1, 2, 3, 4])
    kat["xoshiro_state_1234"] = [str(x.next()) for _ in range(10)]
    for seed in (0, 1, 2, 42, 2**64 - 1):
        x = Xoshiro(seed)
        kat["seed_from_u64"][str(seed)] = {"state": [hex(v) for v in x.s], "first": [str(x.next()) for _ in range(4)]}
    for seed in (0, 1, 7):
        for lo, hi in ((0, 1), (0, 2), (0, 3), (0, 5), (1, 5), (50, 100), (0, 31536000), (0, 2**63), (5, 2**64 - 1)):
            x = Xoshiro(seed)
            vals = [gen_range_attempts(x, lo, hi) for _ in range(6)]
            kat["gen_range"].append({"seed": seed, "lo": str(lo), "hi": str(hi), "values": [str(v) for v, _ in vals],
                                     "attempts": [n for _, n in vals]})
    for lo, hi in ((10**6, 10**7), (0, 1), (1, 2), (9 * 10**8, 11 * 10**8), (9 * 10**8, 21 * 10**8), (10**9, 10**10)):
        mode, low, rg, zone = duration_params(lo, hi)
        kat["duration_params"].append({"lo": str(lo), "hi": str(hi), "mode": mode, "low": str(low), "range": str(rg), "zone": str(zone)})
    json.dump(kat, open(os.path.join(HERE, "rng_kat.json"), "w"), indent=1)

    ex = {}
    for name, w in workloads().items():
        ex[name] = {}
        for cfgname, cfg in (("default", A.Config.default()), ("loss10", A.Config.default(packet_loss_rate=0.1)),
                             ("lat_medium", A.Config.default(lat_lo_ns=9 * 10**8, lat_hi_ns=21 * 10**8))):
            if cfgname != "default" and "pingpong" not in name:
                continue
            ex[name][cfgname] = {str(seed): Sim(w, cfg, seed).run() for seed in (0, 1, 2, 3, 99, 123456789)}
    json.dump(ex, open(os.path.join(HERE, "executor_kat.json"), "w"), indent=1)
    print("wrote rng_kat.json, executor_kat.json")


if __name__ == "__main__":
    main()
