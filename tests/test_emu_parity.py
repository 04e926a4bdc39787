"""Device-code LOGIC vs oracle on CPU: madsim_amd/csrc/sim_kernel.hip compiled for the host with
tests/emu/emu_shim.h (one emulated GPU thread at a time).  This is a debugging aid for GPU-less boxes —
it checks LDS layout arithmetic, heap, mailbox and interpreter; the GPU build itself is checked by
tests/test_gpu_parity.py (-m gpu).  Nothing here is product code or a fallback."""
import random
import sys

import numpy as np
import pytest

import oracle
from madsim_amd import _abi as A
from madsim_amd import workload as W
from tests import emu, fuzz, parity


TALLY = parity.Tally()        # seeds compared / re-run with grown capacities / unresolved, over the whole module (printed at the end)


def _strict(w, seed0, o, e, cfg, lim, what):
    """Every fuzzed seed is a compared seed: a first-pass capacity verdict (MADSIM_OVERFLOW) is re-run with grown capacities
    (tests/parity.py) and what comes back is compared with the oracle like every other seed; none may stay a runner verdict."""
    lim = lim or A.Limits()
    # the expectation is DERIVED from the oracle's run without the workload model's ceilings (parity.expected); the oracle's own
    # model-limits layer — `o`, what the caller ran — must say the same on every fuzzed seed
    want = parity.expected(w, seed0, len(e), cfg, lim)
    assert (want == o).all(), (what, "oracle.run_batch differs from the verdicts derived from oracle.run_batch_pure")
    return parity.compare(e, want, lambda: parity.resolve_seed_by_seed(emu.run_batch, w, seed0, e, cfg, lim),
                          sys._getframe(1).f_code.co_name, TALLY, what, lambda i: parity.beyond_ceiling(w, seed0 + i, cfg, lim))


def _same(w, seed0, n, cfg=None, lim=None):
    o, _ = oracle.run_batch(w, seed0, n, cfg, lim)
    e = emu.run_batch(w, seed0, n, cfg, lim)
    bad = np.nonzero(o != e)[0]
    assert len(bad) == 0, f"seed {seed0 + bad[0]}: emu {e[bad[0]]} != oracle {o[bad[0]]}"
    return o


@pytest.mark.parametrize("nodes,rounds", [(2, 64), (4, 64), (8, 5), (16, 2)])
def test_pingpong(nodes, rounds):
    _same(W.pingpong(nodes, rounds), 0, 400)


@pytest.mark.parametrize("cfg", [A.Config.default(packet_loss_rate=0.05), A.Config.default(buggify=True),
                                 A.Config.default(lat_lo_ns=9 * 10**8, lat_hi_ns=21 * 10**8),
                                 A.Config.default(lat_lo_ns=9 * 10**8, lat_hi_ns=11 * 10**8),
                                 A.Config.default(packet_loss_rate=1.0)])
def test_pingpong_configs(cfg):
    o = _same(W.pingpong(4, 16), 7, 400, cfg)
    if cfg.packet_loss_rate == 1.0:
        assert (o["verdict"] == A.DEADLOCK).all() and (o["msg_count"] == 0).all()


@pytest.mark.parametrize("lw", [8, 16, 32, 64])
def test_lanes_per_wave_invariant(lw):
    lim = A.Limits(); lim.lanes_per_wave = lw
    _same(W.pingpong(4, 8), 0, 300, None, lim)


def test_heap_spill_path():
    """LDS quota of 2 timer entries: everything else lives in the coalesced HBM spill region."""
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 30
    _same(W.pingpong(8, 4), 0, 300, None, lim)


def test_overflow_is_a_verdict_not_a_wrong_answer():
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 0
    e = emu.run_batch(W.pingpong(4, 4), 0, 64, None, lim)
    assert (e["verdict"] == A.OVERFLOW).all()


def test_trace_log_bytes():
    w = W.pingpong(2, 6)
    for seed in (0, 5, 2**40 + 3):
        elog, eres = emu.trace_seed(w, seed)
        olog, ores = oracle.trace_seed(w, seed)
        assert elog == olog and tuple(eres) == ores.astuple()


def test_fuzz_random_workloads():
    """150 random actor programs (bind/send/recv/reply/sleep/yield/clog/set_loss/close, every verdict)."""
    verdicts, n_ovf = set(), 0
    for k in range(150):
        w, cfg, desc = fuzz.random_workload(random.Random(1000 + k))
        lim = fuzz.generous_limits()
        o, _ = oracle.run_batch(w, k * 11, 16, cfg, lim)
        e = emu.run_batch(w, k * 11, 16, cfg, lim)
        n_ovf += int((e["verdict"] == A.OVERFLOW).sum())   # a device capacity verdict: re-run with grown capacities, then compared like every seed
        e = _strict(w, k * 11, o, e, cfg, lim, (k, desc))
        verdicts |= set(o["verdict"].tolist())
    assert {A.PASS, A.DEADLOCK, A.PANIC} <= verdicts and n_ovf < 0.02 * 150 * 16


from tests import lifecycle_workloads as LW  # noqa: E402


def test_trace_build_on_extended_workloads():
    """The trace build (every op class, general addresses): raw log bytes + result on lifecycle / channel / RPC / address workloads."""
    for name in ("kill_restart_with_traffic", "exited", "kv_rpc", "channel_wildcard_listener", "ephemeral_clients",
                 "endpoint_bind_ephemeral", "rpc_hooks", "restart_on_panic_matching", "net_ipless_node"):
        w, cfg, lim = LW.ALL[name](), LW.config(name), LW.limits(name)
        for seed in (0, 3):
            elog, eres = emu.trace_seed(w, seed, cfg, lim)
            olog, ores = oracle.trace_seed(w, seed, cfg, lim)
            assert tuple(eres) == ores.astuple(), (name, seed)
            assert elog == olog, (name, seed)


@pytest.mark.parametrize("name", sorted(LW.ALL))
def test_lifecycle_reference_tests(name):
    """kill / restart / restart_on_panic / pause_resume / exited / join_cancelled ... (task/mod.rs:859-1182)."""
    o = _same(LW.ALL[name](), 0, 128, LW.config(name), LW.limits(name))
    assert (o["verdict"] == (A.PANIC if name in LW.EXPECT_PANIC else A.PASS)).all()


def test_fuzz_lifecycle_workloads():
    """200 random programs with init tasks, kill/restart/pause/resume/abort, restart_on_panic, dead-node spawns."""
    for k in range(200):
        w, cfg, desc = fuzz.random_lifecycle_workload(random.Random(9000 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        o, _ = oracle.run_batch(w, k * 3, 12, cfg, lim)
        e = emu.run_batch(w, k * 3, 12, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_fuzz_guard_workloads():
    """Random lifecycle programs whose task bodies own guards that spawn in Drop (MADSIM_PROG_DROP_SPAWN), in both state layouts."""
    for k in range(120):
        w, cfg, desc = fuzz.random_guard_workload(random.Random(9500 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        if k % 2:
            lim = _global(lim)
        o, _ = oracle.run_batch(w, k * 3, 12, cfg, lim)
        e = emu.run_batch(w, k * 3, 12, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_fuzz_supervisor_workloads():
    """Supervisor calls from every task (spawn / abort / join / kill / restart of own and other nodes), in both state layouts."""
    for k in range(160):
        w, cfg, desc = fuzz.random_supervisor_workload(random.Random(9700 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 48
        if k % 2:
            lim = _global(lim)
        o, _ = oracle.run_batch(w, k * 3, 12, cfg, lim)
        e = emu.run_batch(w, k * 3, 12, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_fuzz_latency_workloads():
    """NetSim::update_config of send_latency from the supervisor and the senders (MS_OP_SET_LATENCY), both state layouts."""
    verdicts = set()
    for k in range(200):
        w, cfg, desc = fuzz.random_latency_workload(random.Random(61000 + k))
        lim = fuzz.mailbox_limits()
        if k % 2:
            lim = _global(lim)
        o, _ = oracle.run_batch(w, k * 5, 12, cfg, lim)
        e = emu.run_batch(w, k * 5, 12, cfg, lim)
        e = _strict(w, k * 5, o, e, cfg, lim, (k, desc,))
        verdicts |= set(o["verdict"].tolist())
    assert {A.PASS, A.DEADLOCK} <= verdicts


def test_fuzz_mixed_workloads():
    """Everything from everywhere: supervisor calls, datagrams, channel and RPC exchanges from every task, both state layouts."""
    for k in range(160):
        w, cfg, desc = fuzz.random_mixed_workload(random.Random(9900 + k))
        lim = fuzz.mixed_limits()
        if k % 2:
            lim = _global(lim)
        if k % 5 >= 3:
            lim.no_trace_hash = 1                      # the reference's plain mode (rand.rs:67), both layouts
        o, _ = oracle.run_batch(w, k * 3, 8, cfg, lim)
        e = emu.run_batch(w, k * 3, 8, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_fuzz_ipvs_workloads():
    """IP Virtual Server rewriting in NetSim::send / connect1 (net/mod.rs:312-317,345-350; net/ipvs.rs), both state layouts."""
    for k in range(160):
        w, cfg, desc = fuzz.random_ipvs_workload(random.Random(9950 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        if k % 2:
            lim = _global(lim)
        o, _ = oracle.run_batch(w, k * 3, 12, cfg, lim)
        e = emu.run_batch(w, k * 3, 12, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_fuzz_ipvs_runtime_workloads():
    """Services changed while the clients run: add_service / del_service / add_server / del_server (net/ipvs.rs:50-85) from
    operator tasks, services that are only declared, calls on a service that is gone (panic) — both state layouts."""
    for k in range(200):
        w, cfg, desc = fuzz.random_ipvs_runtime_workload(random.Random(9975000 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        if k % 2:
            lim = _global(lim)
        o, _ = oracle.run_batch(w, k * 3, 12, cfg, lim)
        e = emu.run_batch(w, k * 3, 12, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_hard_model_limits_are_unsupported_on_both_sides():
    """Limits no `madsim_limits_t` field can lift are MODEL limits: a seventh server of an IPVS service (net/ipvs.rs:66-72 pushes to an
    unbounded Vec), a ninth connection waiting in one accept1 queue (endpoint.rs:307: an unbounded channel), a formatted panic value
    above the workload's declared panic_dyn_max, a 255th live task (task/mod.rs:607-654: an unbounded set), a 256th registration of one
    socket (endpoint.rs:288-300: an unbounded Vec), a 16th payload queued in one channel direction (net/mod.rs:417-421: an unbounded channel).  Kernel and oracle give MADSIM_UNSUPPORTED at the same instruction, every other
    field 0 — never MADSIM_OVERFLOW (a re-run could not resolve it), never a shorter list."""
    ws = [(w, lim) for _, w, lim, _ in LW.model_ceiling_workloads()]
    for w, lim in ws:
        o, _ = oracle.run_batch(w, 0, 8, None, lim)
        assert (o["verdict"] == A.UNSUPPORTED).all() and not o["steps"].any() and not o["rng_calls"].any()
        for glob in (0, 1):
            e = emu.run_batch(w, 0, 8, None, _global(lim) if glob else lim)
            assert (e == o).all(), (glob, e[0], o[0])


def test_baseline_config_shaped_workloads():
    """configs[2]-shaped election loop (timeouts, partitions, HBM heap spill) and configs[3]-shaped KV-RPC."""
    o = _same(W.raft_election(), 0, 300, None, W.raft_election_limits())
    assert (o["verdict"] == A.PASS).all()
    big = W.raft_election_limits(); big.mbox_regs, big.mbox_msgs = 96, 16      # loss: more timeouts, more dead registrations
    o = _same(W.raft_election(), 0, 300, A.Config.default(packet_loss_rate=0.05), big)
    o = _same(W.kv_rpc(), 0, 300, None, W.kv_rpc_limits())
    assert (o["verdict"] == A.PASS).all()
    o = _same(W.streaming_topology(), 0, 96, None, W.streaming_topology_limits())        # configs[4] shape
    assert (o["verdict"] == A.PASS).all()
    o = _same(W.timer_storm(), 0, 200, None, W.timer_storm_limits(4))      # base ops on the runtime-lane-stride build, heap mostly spilled
    assert (o["verdict"] == A.PASS).all()


def test_every_bench_workload_runs_on_the_build_its_ops_need():
    """select_variant: single-class workloads get the pruned builds (timeouts only / channel only), base-op workloads never
    pay for extended ops, mixed workloads take the full build — without general address resolution (FEAT 15) when every
    address is a plain node IP and the state lives in global memory."""
    from madsim_amd import runtime
    want = {"pingpong": (0, 6, 0), "timers": (0, 15, 0), "raft": (1, 6, 16), "kv": (2, 6, 16), "topo": (15, 6, 16)}   # (raft: full waves again since round 6 — the narrow heap entries)
    for name, (feat, lws, glob) in want.items():
        w, lim, _ = W.bench_case(name)
        g = runtime.geometry(w, lim)
        assert ((g.variant >> 8) & 0x1f, (g.variant >> 16) & 0xf, g.variant & 16) == (feat, lws, glob), (name, hex(g.variant))
        assert (g.global_bytes_per_seed > 0) == bool(glob) and (not glob or g.lanes_per_wave == 64)
        assert bool((g.variant >> 8) & 0x80) == (name in ("raft", "topo"))            # 8-byte heap entries (MADSIM_STATE_NARROW_HEAP) for the two spilling workloads
        lim.state_mem = A.STATE_LDS                     # the LDS-resident layout stays selectable
        g = runtime.geometry(w, lim)
        assert g.variant & 16 == 0 and g.global_bytes_per_seed == 0


def test_fuzz_rpc_workloads():
    """200 random typed-RPC programs (net/rpc.rs): call / call_timeout against handler tasks, slow and silent handlers,
    loss, server kill/restart."""
    for k in range(200):
        w, cfg, desc = fuzz.random_rpc_workload(random.Random(31000 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        o, _ = oracle.run_batch(w, k * 5, 12, cfg, lim)
        e = emu.run_batch(w, k * 5, 12, cfg, lim)
        e = _strict(w, k * 5, o, e, cfg, lim, (k, desc,))
        assert (e["verdict"] == A.OVERFLOW).mean() < 0.1


def test_fuzz_address_resolution():
    """200 random programs over node-IP / 0.0.0.0 / 127.0.0.1 addresses, IP-less nodes, duplicate and unbound entries."""
    verdicts = set()
    for k in range(200):
        w, cfg, desc = fuzz.random_addr_workload(random.Random(73000 + k))
        lim = fuzz.generous_limits()
        o, _ = oracle.run_batch(w, k * 9, 12, cfg, lim)
        e = emu.run_batch(w, k * 9, 12, cfg, lim)
        e = _strict(w, k * 9, o, e, cfg, lim, (k, desc,))
        verdicts |= set(o["verdict"].tolist())
    assert A.PASS in verdicts and A.PANIC in verdicts


def test_fuzz_ephemeral_ports():
    """200 random programs binding port 0 (network.rs:224-236): the oracle hands out ports literally (lowest free port of
    the node for that IP, the sender's port captured in `from`), the kernel binds the first free candidate entry of the
    handle (geometry.h device_socks) — they have to agree on every port observed and on everything downstream."""
    n_overflow = 0
    for k in range(200):
        w, cfg, desc = fuzz.random_ephemeral_workload(random.Random(74000 + k))
        lim = fuzz.generous_limits()
        if k % 2:
            lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL
        o, _ = oracle.run_batch(w, k * 7, 12, cfg, lim)
        e = emu.run_batch(w, k * 7, 12, cfg, lim)
        e = _strict(w, k * 7, o, e, cfg, lim, (k, desc,))
        n_overflow += int((e["verdict"] == A.OVERFLOW).sum())
    assert n_overflow == 0                          # one live Endpoint per entry: a candidate is always free


def test_fuzz_channel_guards():
    """250 random reliable-channel programs about who keeps an address bound (every Sender / Receiver holds a clone of its
    Endpoint's Arc<BindGuard>, endpoint.rs:181-210): listeners dropped under live connections, clients dropping their
    Endpoint before their connection, probers binding the same address, ephemeral and wildcard addresses, clogged links."""
    seen, n_ovf = set(), 0
    for k in range(250):
        w, cfg, desc = fuzz.random_channel_workload(random.Random(75000 + k))
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        if k % 2:
            lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL
        o, _ = oracle.run_batch(w, k * 5, 10, cfg, lim)
        e = emu.run_batch(w, k * 5, 10, cfg, lim)
        e = _strict(w, k * 5, o, e, cfg, lim, (k, desc,))
        n_ovf += int((e["verdict"] == A.OVERFLOW).sum())
        seen |= set(o["verdict"].tolist())
    assert A.PASS in seen and A.DEADLOCK in seen
    assert n_ovf < 0.05 * 250 * 10, n_ovf            # (more than eight connections waiting in one accept queue: a device capacity)


def test_fuzz_rpc_hooks_and_panic_codes():
    """Random typed-RPC programs with NetSim request / response hooks installed and replaced at random moments."""
    for k in range(150):
        w, cfg, desc = fuzz.random_rpc_workload(random.Random(61000 + k), hooks=True)
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        o, _ = oracle.run_batch(w, k * 5, 12, cfg, lim)
        e = emu.run_batch(w, k * 5, 12, cfg, lim)
        e = _strict(w, k * 5, o, e, cfg, lim, (k, desc,))


def _global(lim=None):
    """The same limits with the task table and planes forced into the per-lane global-memory block (Variant::G)."""
    g = A.Limits()
    if lim is not None:
        for f, _ in A.Limits._fields_:
            setattr(g, f, getattr(lim, f))
    g.lanes_per_wave, g.state_mem = 0, A.STATE_GLOBAL
    return g


@pytest.mark.parametrize("name", sorted(LW.ALL))
def test_global_state_layout_lifecycle(name):
    """Every reference lifecycle / channel / RPC test again with per-seed state in global memory instead of LDS."""
    o = _same(LW.ALL[name](), 0, 96, LW.config(name), _global(LW.limits(name)))
    assert (o["verdict"] == (A.PANIC if name in LW.EXPECT_PANIC else A.PASS)).all()


def test_global_state_layout_fuzz():
    for k in range(120):
        gen = fuzz.random_lifecycle_workload if k % 2 else fuzz.random_rpc_workload
        w, cfg, desc = gen(random.Random(47000 + k))
        lim = _global(fuzz.generous_limits()); lim.max_tasks = 24
        o, _ = oracle.run_batch(w, k * 7, 12, cfg, lim)
        e = emu.run_batch(w, k * 7, 12, cfg, lim)
        e = _strict(w, k * 7, o, e, cfg, lim, (k, desc,))


def test_global_state_is_chosen_by_footprint_and_can_be_forced_either_way():
    from madsim_amd import runtime
    small = LW.ALL["kill"]()                       # extended ops, tiny state: stays in LDS on full waves
    assert runtime.geometry(small).variant & 16 == 0
    assert runtime.geometry(small, _global()).variant & 16
    lim = A.Limits(); lim.state_mem = A.STATE_GLOBAL
    assert runtime.geometry(W.pingpong(4, 4), lim).variant & 16 == 0      # base-op workloads have no global-state build
    lim = A.Limits(); lim.state_mem = 3
    with pytest.raises(runtime.MadsimHipError, match="state_mem"):
        runtime.geometry(small, lim)


def _without_trace_hash(lim):
    lim = lim or A.Limits()
    lim.no_trace_hash = 1
    return lim


@pytest.mark.parametrize("case", ["pingpong", "pingpong_rq", "spill", "lanes16", "raft", "kv", "topo", "timers"])
def test_no_trace_hash_drops_only_the_fingerprint(case):
    """madsim_limits_t.no_trace_hash: the reference's plain run (rand.rs:67: no log, no check).  trace_hash comes back 0, every
    other field is what the logging run reports — on the build compiled without the fold (base ops, full waves) and on the
    builds that test the flag at run time; a trace request logs regardless."""
    lim = None
    if case == "pingpong":
        w = W.pingpong(4, 16)
    elif case == "pingpong_rq":
        w, lim, _ = W.bench_case("pingpong", 4, 16, 4)
    elif case == "spill":
        w = W.pingpong(8, 4); lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 30
    elif case == "lanes16":
        w = W.pingpong(4, 8); lim = A.Limits(); lim.lanes_per_wave = 16
    else:
        w, lim, _ = W.bench_case(case, 4, 8, 4)
    n = 64 if case in ("raft", "topo") else 192
    with_log = _same(w, 3, n, None, lim)
    lim2 = _without_trace_hash(lim)
    without = _same(w, 3, n, None, lim2)
    assert (without["trace_hash"] == 0).all() and (with_log["trace_hash"] != 0).all()
    for f in ("verdict", "steps", "clock_ns", "msg_count", "rng_calls", "obs_hash"):
        assert (without[f] == with_log[f]).all(), f
    elog, eres = emu.trace_seed(w, 3, None, lim2)
    olog, ores = oracle.trace_seed(w, 3, None, lim2)
    assert bytes(elog) == bytes(olog) and len(olog) > 0
    assert tuple(eres) == ores.astuple() and ores.trace_hash == with_log["trace_hash"][0]


def _compact(lim):
    lim.state_mem = A.STATE_COMPACT
    return lim


@pytest.mark.parametrize("nodes,rounds,loss", [(4, 64, 0.0), (2, 64, 0.0), (6, 9, 0.0), (4, 16, 0.05), (4, 16, 1.0)])
def test_compact_layout_pingpong(nodes, rounds, loss):
    """MADSIM_STATE_COMPACT: 8-byte heap entries (low deadline word), heap root in registers, main task in global memory — the
    same answers as the oracle on every field, with and without the determinism-log fold."""
    w, lim, _ = W.bench_case("pingpong", nodes, rounds, 4)
    lim.heap_lds_slots = max(4, nodes)
    cfg = A.Config.default(packet_loss_rate=loss) if loss else None
    _same(w, 5, 320, cfg, _compact(lim))
    lim.no_trace_hash = 1
    _same(w, 5, 320, cfg, lim)
    g = emu.geometry(w, lim)
    assert g.lds_bytes_per_seed == (lim.heap_lds_slots - 1) * 8 + nodes * 24 + nodes * 8


def test_compact_layout_is_automatic_when_it_gains_a_workgroup_and_refused_outside_its_horizon():
    w, lim, _ = W.bench_case("pingpong", 4, 64, 4)
    assert emu.geometry(w, lim).lds_bytes_per_seed == 152          # auto: 200 -> 152 bytes, four workgroups per CU instead of three
    lim.state_mem = A.STATE_LDS
    assert emu.geometry(w, lim).lds_bytes_per_seed == 200          # the plain layout stays selectable
    # a sleep beyond the 2^31 ns horizon: forced compact is refused, auto falls back to the plain layout
    wl = W.WorkloadBuilder()
    n1 = wl.create_node()
    t = wl.task(n1); t.sleep(secs=3)
    m = wl.main(); m.spawn(t); m.join(t)
    far = wl.build()
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 4, 0
    _same(far, 0, 64, None, lim)
    with pytest.raises(RuntimeError):
        emu.run_batch(far, 0, 64, None, _compact(lim))
    # buggify's 1..5 s delays rule it out as well
    w, lim, _ = W.bench_case("pingpong", 4, 8, 4)
    with pytest.raises(RuntimeError):
        emu.run_batch(w, 0, 64, A.Config.default(buggify=True), _compact(lim))
    lim.state_mem = A.STATE_AUTO
    _same(w, 0, 192, A.Config.default(buggify=True), lim)
    assert emu.geometry(w, lim).lds_bytes_per_seed == 152          # (geometry alone does not know the config: the run decides)


def test_compact_layout_fuzz():
    """Random base-op programs small enough for the compact layout (<= 8 tasks, sleeps below 2.1 s), compact vs oracle."""
    ran = 0
    for k in range(400):
        w, cfg, desc = fuzz.random_workload(random.Random(31000 + k))
        lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 24, 0
        lim.mbox_regs, lim.mbox_msgs = 4, 6
        _compact(lim)
        try:
            e = emu.run_batch(w, k * 5, 24, cfg, lim)
        except RuntimeError:
            continue                                               # not a compact candidate (tasks, horizon, buggify)
        o, _ = oracle.run_batch(w, k * 5, 24, cfg, lim)
        e = _strict(w, k * 5, o, e, cfg, lim, (k, desc,))
        ran += 1
    assert ran >= 60, ran


@pytest.mark.parametrize("sched", [A.SCHED_STATIC, A.SCHED_QUEUE])
def test_compact_layout_lane_reuse(sched):
    """More seeds than lanes: a lane runs several seeds in turn — the main task's global record, the register-resident heap
    root and the biased LDS task arrays are all re-initialised by seed_init."""
    w, lim, _ = W.bench_case("pingpong", 4, 8, 4)
    lim.sched = sched
    o, _ = oracle.run_batch(w, 9, 2600, A.Config.default(packet_loss_rate=0.02), lim)
    e = emu.run_batch(w, 9, 2600, A.Config.default(packet_loss_rate=0.02), lim, num_cus=1)
    assert emu.geometry(w, lim).lds_bytes_per_seed == 152 and (o == e).all()


# ---- MADSIM_STATE_DEDUP_TIMERS: re-registered Sleep timers as counts (k_timer.h dedup_note) ---------------------------------

def test_dedup_timers_switch_selects_the_timeout_only_global_build_and_nothing_else():
    raft = W.raft_election()
    on = emu.geometry_params(raft, LW.dedup_limits(W.raft_election_limits()))
    assert on["features"] == 1 and on["gstate_mode"] == 1 and on["dedup_n"] == 64          # MADSIM_FEAT_TIME only, global state
    gran = 32
    while gran < on["task_units"] * 16:
        gran *= 2                    # a task slot's units share one power-of-two granule per lane (k_state.h gs_addr_task)
    assert on["dedup_off"] == on["max_tasks"] * gran and on["gs_planes"] == on["dedup_off"] + 64 * 16
    lim = W.raft_election_limits(); lim.state_mem = A.STATE_GLOBAL
    off = emu.geometry_params(raft, lim)
    assert off["dedup_n"] == 0 and off["gs_planes"] == on["gs_planes"] - 64 * 16
    lim = W.raft_election_limits(); lim.state_mem = A.STATE_LDS | A.STATE_DEDUP_TIMERS        # LDS-resident: ignored
    assert emu.geometry_params(raft, lim)["dedup_n"] == 0
    for w, base in ((W.kv_rpc(), W.kv_rpc_limits()), (W.streaming_topology(), W.streaming_topology_limits()), (W.pingpong(4, 8), None)):
        assert emu.geometry_params(w, LW.dedup_limits(base))["dedup_n"] == 0                  # other op classes / base ops: ignored
    lim = A.Limits(); lim.state_mem = 0x400
    with pytest.raises(RuntimeError, match="state_mem"):
        emu.geometry_params(raft, lim)


def test_dedup_timers_election_loop():
    """The election loop (a third of its Timer::add calls re-register a pending Sleep) with those calls kept as counts:
    every result byte as the oracle has it, in static striding and through the work queue (several seeds per lane)."""
    w, lim = W.raft_election(), LW.dedup_limits(W.raft_election_limits())
    o = _same(w, 5000, 192, None, lim)
    assert (o["verdict"] == A.PASS).mean() > 0.9
    want, _ = oracle.run_batch(w, 0, 2000, None, lim)
    for sched in (A.SCHED_STATIC, A.SCHED_QUEUE):               # one CU: 768 lanes, so a lane runs two or three seeds in turn
        lim.sched = sched
        assert (emu.run_batch(w, 0, 2000, None, lim, num_cus=1) == want).all()


def test_dedup_timers_ties_restart_the_seed_with_the_literal_heap():
    """Two DIFFERENT events on one deadline fire in the order of the BinaryHeap's array, which a heap without the repeats does
    not share: timeout_repeats_and_ties has such ties in a seventh of its seeds (with the tie check compiled out 9 of these
    512 seeds differ from the oracle — tried once by hand); the kernel notices each as it pops it and runs the seed again
    with every timer a heap entry."""
    w = LW.timeout_repeats_and_ties()
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 4, 60; lim.mbox_regs, lim.mbox_msgs = 8, 8
    o = _same(w, 0, 512, None, LW.dedup_limits(lim))
    assert (o["verdict"] == A.PASS).all()
    assert emu.geometry_params(w, LW.dedup_limits(lim))["dedup_n"] == 64
    _same(w, 0, 128, A.Config.default(packet_loss_rate=0.1), LW.dedup_limits(lim))
    # a lane that ran a seed twice goes back to counting for its next seed (one CU, several seeds per lane, both work distributions)
    want, _ = oracle.run_batch(w, 700, 2400, None, lim)
    for sched in (A.SCHED_STATIC, A.SCHED_QUEUE):
        l2 = LW.dedup_limits(lim); l2.sched = sched
        assert (emu.run_batch(w, 700, 2400, None, l2, num_cus=1) == want).all()


def test_dedup_timers_fuzz():
    """Random timeout-only programs (repeats, nanosecond ties, sleep_until / advance, partitions) on the de-duplicating build."""
    seen, active = set(), 0
    for k in range(250):
        w, cfg, desc = fuzz.random_timeout_workload(random.Random(92000 + k))
        lim = LW.dedup_limits(fuzz.generous_limits())
        if k % 3 == 2:
            lim.no_trace_hash = 1
        o, _ = oracle.run_batch(w, k * 7, 12, cfg, lim)
        e = emu.run_batch(w, k * 7, 12, cfg, lim)
        e = _strict(w, k * 7, o, e, cfg, lim, (k, desc,))
        seen |= set(o["verdict"].tolist())
        active += emu.geometry_params(w, lim)["dedup_n"] != 0
    assert {A.PASS, A.PANIC, A.DEADLOCK} <= seen and active > 200


# ---- MADSIM_STATE_NARROW_HEAP: 8-byte timer-heap entries + delivery record pool (k_timer.h nh_*, round 6) -------------------------

def _narrow(lim, heap_lds=None):
    import copy
    l2 = copy.copy(lim) if lim is not None else A.Limits()
    l2.state_mem = (l2.state_mem if (l2.state_mem & 0xff) else A.STATE_GLOBAL) | A.STATE_NARROW_HEAP
    if l2.lanes_per_wave != 32:
        l2.lanes_per_wave = 0                   # (the fuzz limits ask for sub-wave occupancy of the LDS-resident builds)
    if heap_lds is not None:
        l2.heap_spill_slots, l2.heap_lds_slots = l2.heap_spill_slots + max(0, l2.heap_lds_slots - heap_lds), heap_lds
    return l2


def test_narrow_heap_switch_selects_builds_with_a_spill_region_and_a_short_horizon():
    topo, raft = W.streaming_topology(), W.raft_election()
    g = emu.geometry_params(topo, W.streaming_topology_limits())                           # (the bench case asks for the narrow entries itself)
    assert g["narrow"] == 1 and g["pool_n"] == 64 and g["heap_lds"] == 31 and g["lds_per_seed"] == 31 * 8 + 48
    lim = W.streaming_topology_limits(); lim.state_mem &= ~A.STATE_NARROW_HEAP
    wide = emu.geometry_params(topo, lim)
    assert wide["narrow"] == 0 and wide["heap_lds"] == 16 and wide["lds_per_seed"] == g["lds_per_seed"]     # the same LDS: 16 entries of 16 bytes
    assert g["gs_stride"] == wide["gs_stride"] + 64 * 8                                                     # the record pool behind the planes
    g = emu.geometry_params(raft, W.raft_election_limits())                                # (full waves: what fits beside three waves per SIMD)
    assert g["narrow"] == 1 and g["heap_lds"] == 20 and g["dedup_n"] == 64                 # with the re-registration counts
    l32 = W.raft_election_limits(); l32.lanes_per_wave = 32
    assert emu.geometry_params(raft, _narrow(l32, 44))["heap_lds"] == 44                   # 32 seed lanes per wave: twice the LDS per seed
    # not for: LDS-resident state, connection-only workloads (no build), base ops, a workload that sleeps past 2^31 ns, buggify
    lim = W.raft_election_limits(); lim.state_mem = A.STATE_LDS | A.STATE_NARROW_HEAP; lim.lanes_per_wave = 0
    assert emu.geometry_params(raft, lim)["narrow"] == 0
    assert emu.geometry_params(W.kv_rpc(), _narrow(W.kv_rpc_limits()))["narrow"] == 0
    assert emu.geometry_params(W.pingpong(4, 8), _narrow(None))["narrow"] == 0
    wl = W.WorkloadBuilder(); n = wl.create_node(); a = wl.addr(n, 1)
    t = wl.task(n); t.bind(a); t.recv_from_timeout(a, 1, ms=5); t.sleep(secs=3)
    m = wl.main(); m.spawn(t); m.join(t)
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 30
    assert emu.geometry_params(wl.build(), _narrow(lim))["narrow"] == 0


def test_narrow_heap_topology_and_election_loop():
    """configs[2] / configs[4] shapes on 8-byte heap entries: every result byte as the oracle has it, at several LDS quotas (the whole
    heap spilled below the root's children .. the whole heap in LDS), static striding and work queue with several seeds per lane."""
    for w, lim, quotas in ((W.streaming_topology(), W.streaming_topology_limits(), (3, 15, 31, 120)),
                           (W.raft_election(), W.raft_election_limits(), (2, 22, 44))):
        want, _ = oracle.run_batch(w, 0, 900, None, lim)
        for q in quotas:
            l2 = _narrow(lim, q)
            assert emu.geometry_params(w, l2)["narrow"] == 1
            for sched in (A.SCHED_STATIC, A.SCHED_QUEUE):
                l2.sched = sched
                e = emu.run_batch(w, 0, 900 if q == quotas[1] else 200, None, l2, num_cus=1)
                assert (e == want[:len(e)]).all(), (q, sched)
        _same(w, 77, 96, A.Config.default(packet_loss_rate=0.05), _narrow(lim, quotas[1]))


def test_narrow_heap_horizon_is_a_capacity_verdict_and_the_rerun_is_wide():
    """A deadline 2^31 ns or more ahead of the clock cannot live in an 8-byte entry: the host refuses the layout when the workload can
    ask for one by itself (above); what only shows at run time — a channel back-off that doubled past 2 s under a long partition
    (net/mod.rs:388-398) — is a capacity verdict on that seed, and the re-run (tests/parity.py grow = madsim_hip.cpp grow) leaves the
    narrow layout and answers what the oracle answers."""
    w, lim = LW.narrow_heap_backoff_past_the_horizon()
    l2 = _narrow(lim)
    assert emu.geometry_params(w, l2)["narrow"] == 1
    o, _ = oracle.run_batch(w, 0, 64, None, l2)
    e = emu.run_batch(w, 0, 64, None, l2)
    assert (o["verdict"] == A.PASS).all() and (e["verdict"] == A.OVERFLOW).all()           # the 4 096 ms back-off
    _strict(w, 0, o, e, None, l2, "narrow horizon")


def test_narrow_heap_pool_exhaustion_is_a_capacity_verdict():
    """More datagrams in flight than the record pool holds (32 with a small heap): MADSIM_OVERFLOW, re-run, compared."""
    w, cfg, lim = LW.narrow_heap_forty_datagrams_in_flight()
    l2 = _narrow(lim)
    g = emu.geometry_params(w, l2)
    assert g["narrow"] == 1 and g["pool_n"] == 32
    o, _ = oracle.run_batch(w, 0, 48, cfg, l2)
    e = emu.run_batch(w, 0, 48, cfg, l2)
    assert (e["verdict"] == A.OVERFLOW).any()
    _strict(w, 0, o, e, cfg, l2, "narrow pool")


def test_narrow_heap_fuzz():
    """Random programs of the generators whose builds carry the variant — timeout-only (with and without the re-registration counts,
    64 and 32 seed lanes per wave), latency switches, mixed / supervisor / RPC (every op class, plain addresses) — on 8-byte entries
    with small LDS quotas, so pushes and pops walk the spilled levels; deliveries ride the record pool."""
    seen, active = set(), 0
    gens = [fuzz.random_timeout_workload, fuzz.random_latency_workload, fuzz.random_mixed_workload, fuzz.random_rpc_workload,
            fuzz.random_timeout_workload, fuzz.random_workload]
    for k in range(360):
        gen = gens[k % len(gens)]
        w, cfg, desc = gen(random.Random(66000 + k))[:3]
        lim = fuzz.mixed_limits() if gen is fuzz.random_mixed_workload else fuzz.mailbox_limits()
        lim = _narrow(lim, [1, 2, 3, 5, 8][k % 5])
        if gen is fuzz.random_timeout_workload and k % 2:
            lim.state_mem |= A.STATE_DEDUP_TIMERS
            if k % 4 == 3:
                lim.lanes_per_wave = 32
        if k % 7 == 6:
            lim.no_trace_hash = 1
        try:
            e = emu.run_batch(w, k * 7, 10, cfg, lim)
        except RuntimeError:
            continue
        o, _ = oracle.run_batch(w, k * 7, 10, cfg, lim)
        e = _strict(w, k * 7, o, e, cfg, lim, (k, gen.__name__, desc,))
        seen |= set(o["verdict"].tolist())
        active += emu.geometry_params(w, lim)["narrow"] != 0
    assert {A.PASS, A.PANIC, A.DEADLOCK} <= seen and active > 150, active


def test_fuzz_reply_without_receive():
    """`reply` before anything was received / after a timed-out receive: an unset `from` is socket-table entry 0 on both sides
    (found by the timeout generator's first version: the oracle had sent such a reply to port 0, the kernel to entry 0's port)."""
    for k in range(150):
        w, cfg, desc = fuzz.random_reply_without_receive_workload(random.Random(96000 + k))
        lim = fuzz.mailbox_limits() if k % 4 < 3 else fuzz.generous_limits()      # (every fourth program at 15 messages: the re-run path)
        if k % 2:
            lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL
        o, _ = oracle.run_batch(w, k * 3, 8, cfg, lim)
        e = emu.run_batch(w, k * 3, 8, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, desc,))


def test_fuzz_unstructured_workloads():
    """Op soup (tests/fuzz.py random_unstructured_workload): whatever validate() lets through gets the oracle's answer — or is
    refused by both layouts alike."""
    from madsim_amd import runtime
    seen, refused = set(), 0
    for k in range(400):
        w, cfg, desc = fuzz.random_unstructured_workload(random.Random(97000 + k))
        for glob in (0, 1):
            lim = fuzz.generous_limits(); lim.max_tasks = 16
            if glob:
                lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS
            try:
                e = emu.run_batch(w, k * 3, 6, cfg, lim)
            except RuntimeError as ex:                     # a validate() rule (e.g. two listening Endpoints on a node that can be killed)
                assert "emu error" in str(ex), ex
                refused += 1
                continue
            o, _ = oracle.run_batch(w, k * 3, 6, cfg, lim)
            e = _strict(w, k * 3, o, e, cfg, lim, (k, glob, desc,))
            seen |= set(o["verdict"].tolist())
    assert {A.PASS, A.PANIC, A.DEADLOCK} <= seen and refused < 40


def test_t0_must_be_marked_before_it_is_read_and_cclose_alone_lays_out_the_connection_unit():
    wl = W.WorkloadBuilder(); n = wl.create_node()
    t = wl.task(n); t.sleep_until(ms=1); t.mark(); t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    with pytest.raises(RuntimeError, match="t0 is not assigned"):
        emu.run_batch(wl.build(), 0, 1)
    wl = W.WorkloadBuilder(); n = wl.create_node()
    t = wl.task(n); t.mark(); t.assert_elapsed("<", ms=1); t.chan_close(); t.done()      # a stray drop((tx, rx)): no connection in hand
    m = wl.main(); m.spawn(t); m.join(t)
    w = wl.build()
    assert emu.geometry_params(w)["features"] & 2                                      # MADSIM_FEAT_CHAN: the connection unit exists
    o = _same(w, 0, 8)
    assert (o["verdict"] == A.PASS).all()


def test_fuzz_unstructured_workloads_under_varied_configs_and_limits():
    """The op soup again with what a run can be given besides the program: buggify, second-scale and nanosecond latency ranges,
    heavy loss; sub-wave lane strides, the global-state layout with the timer switch, a two-entry LDS heap (everything else in
    the spill region), time limits, step caps, no log fingerprint — verdicts TIME_LIMIT and STEP_LIMIT included."""
    seen = set()
    for k in range(600):
        w, _, desc = fuzz.random_unstructured_workload(random.Random(98000 + k))
        lr = random.Random(k * 7 + 1)
        lo, hi = lr.choice([(1_000_000, 10_000_000), (1, 2), (0, 10_000_000), (900_000_000, 2_100_000_000), (5_000_000, 5_000_001)])
        cfg = A.Config.default(packet_loss_rate=lr.choice([0.0, 0.05, 0.5]), lat_lo_ns=lo, lat_hi_ns=hi, buggify=lr.random() < 0.3,
                               loss_table=(0.0, 0.5, 1.0))
        lim = fuzz.generous_limits(); lim.max_tasks = 16
        mode = lr.randrange(6)
        if mode == 0: lim.lanes_per_wave = lr.choice([8, 16, 32])
        elif mode == 1: lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS
        elif mode == 2: lim.heap_lds_slots, lim.heap_spill_slots = 2, 62
        elif mode == 3: lim.time_limit_ns = lr.choice([1, 1_000_000, 5_000_000, 2_000_000_000])
        elif mode == 4: lim.max_steps = lr.choice([1, 7, 40, 200])
        else: lim.no_trace_hash = 1; lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL
        try:
            e = emu.run_batch(w, k * 3, 6, cfg, lim)
        except RuntimeError:                               # refused by validate()
            continue
        o, _ = oracle.run_batch(w, k * 3, 6, cfg, lim)
        e = _strict(w, k * 3, o, e, cfg, lim, (k, mode, desc,))
        seen |= set(o["verdict"].tolist())
    assert {A.PASS, A.PANIC, A.DEADLOCK, A.TIME_LIMIT, A.STEP_LIMIT} <= seen


def test_fuzz_unstructured_wide_workloads():
    """The op soup over the whole table format (tests/fuzz.py random_unstructured_wide_workload), both state layouts: the oracle's
    answer on all 48 bytes, a capacity verdict, or a refusal by validate() — and MADSIM_UNSUPPORTED exactly where the oracle says
    it (a port-0 entry bound again beside its live Endpoint).  No seed ends MADSIM_INTERNAL."""
    seen, refused = set(), 0
    for k in range(500):
        w, cfg, desc = fuzz.random_unstructured_wide_workload(random.Random(1_500_000 + k))
        for glob in (0, 1):
            lim = fuzz.wide_limits(glob)
            try:
                e = emu.run_batch(w, k * 3, 6, cfg, lim)
            except RuntimeError as ex:
                assert "emu error" in str(ex), ex
                refused += 1
                break
            o, _ = oracle.run_batch(w, k * 3, 6, cfg, lim)
            e = _strict(w, k * 3, o, e, cfg, lim, (k, glob, desc,))
            assert not (e["verdict"] == A.INTERNAL).any()
            seen |= set(o["verdict"].tolist())
    assert {A.PASS, A.PANIC, A.DEADLOCK, A.UNSUPPORTED} <= seen and refused < 25


def test_port0_entry_bound_again_beside_its_live_endpoint_is_unsupported():
    """An entry names one Endpoint at a time.  `bind; bind` on a port-0 entry = two Endpoints under one name: MADSIM_UNSUPPORTED
    with every other field 0, from the oracle and the kernel alike; `bind; close; bind` is the reference's `bind` test shape and passes."""
    for glob in (0, 1):
        wl = W.WorkloadBuilder(); n = wl.create_node()
        a = wl.addr(n, 0, ip="unspecified")
        t = wl.task(n); t.bind(a); t.bind(a); t.done()
        m = wl.main(); m.spawn(t); m.join(t); m.done()
        w = wl.build()
        lim = fuzz.wide_limits(glob)
        o, _ = oracle.run_batch(w, 0, 8, None, lim)
        e = emu.run_batch(w, 0, 8, None, lim)
        assert (o == e).all() and (o["verdict"] == A.UNSUPPORTED).all()
        assert not o["steps"].any() and not o["clock_ns"].any() and not o["trace_hash"].any() and not o["rng_calls"].any()
        wl = W.WorkloadBuilder(); n = wl.create_node()
        a = wl.addr(n, 0, ip="unspecified")
        t = wl.task(n); t.bind(a); t.close(a); t.bind(a); t.done()
        m = wl.main(); m.spawn(t); m.join(t); m.done()
        w = wl.build()
        o, _ = oracle.run_batch(w, 0, 8, None, lim)
        e = emu.run_batch(w, 0, 8, None, lim)
        assert (o == e).all() and (o["verdict"] == A.PASS).all()


def test_socket_of_a_restarted_nodes_dead_task_serves_a_receive_another_holder_registered():
    """TaskHandle::restart resets no sockets and BindGuard::drop returns early on a killed node (net/mod.rs:483-493,
    task/mod.rs:374-401): the dead task's socket stays in the table.  A datagram for it wakes a receive that a task of ANOTHER node
    registered on that entry before the restart (an Endpoint clone held across it) and is dropped when nobody did — in kernel
    builds with and without connection ops alike (round 3: they disagreed with each other)."""
    outs = []
    for with_conn in (False, True):
        wl = W.WorkloadBuilder()
        n1, n2 = wl.create_node(), wl.create_node()
        a1, a2 = wl.addr(n1, 1), wl.addr(n2, 1)
        binder = wl.task(n1); binder.bind(a1); binder.sleep(ms=100); binder.done()
        holder = wl.task(n2); holder.mark(); holder.sleep(ms=2); holder.recv_from_timeout(a1, 1, ms=50); holder.trace_val(); holder.done()
        sender = wl.task(n2); sender.bind(a2); sender.sleep(ms=10); sender.send_to(a2, a1, 1, 77); sender.sleep(ms=20)
        sender.send_to(a2, a1, 1, 78); sender.sleep(ms=20)
        if with_conn:
            sender.chan_close()
        sender.done()
        sup = wl.task(n2); sup.sleep(ms=5); sup.restart(n1); sup.done()
        m = wl.main()
        for t in (binder, holder, sender, sup):
            m.spawn(t)
        for t in (holder, sender, sup):
            m.join(t)
        m.done()
        w = wl.build()
        for glob in (0, 1):
            lim = fuzz.wide_limits(glob)
            o, _ = oracle.run_batch(w, 0, 16, None, lim)
            e = emu.run_batch(w, 0, 16, None, lim)
            assert (o == e).all(), (with_conn, glob, o[o != e][0], e[o != e][0])
            outs.append(o)
    # the connection op changes the kernel build, not the run: same verdicts, clocks and observations
    for f in ("verdict", "steps", "clock_ns", "msg_count", "rng_calls", "trace_hash", "obs_hash"):
        assert (outs[0][f] == outs[2][f]).all(), f


@pytest.mark.parametrize("name", ["receiver_drop", "request_timeout_with_stale_timers", "dead_registrations_swept_by_delivery",
                                  "many_endpoints_dropped_in_table_order"])
def test_global_state_with_32_seed_lanes_per_wave(name):
    """Round 4: the timeout-only global-state build also runs 32 seed lanes per wave (the election loop's bench case): same bytes."""
    lim = LW.limits(name) or A.Limits()
    lim.lanes_per_wave, lim.state_mem = 32, A.STATE_GLOBAL
    g = emu.geometry_params(LW.ALL[name](), lim)
    assert g["gstate_mode"] == 1 and g["features"] == 1
    _same(LW.ALL[name](), 0, 160, LW.config(name), lim)
    lim.state_mem |= A.STATE_DEDUP_TIMERS
    _same(LW.ALL[name](), 0, 160, A.Config.default(packet_loss_rate=0.05), lim)


def test_global_state_32_lanes_is_refused_for_other_op_classes():
    lim = LW.limits("kv_rpc") or A.Limits()
    lim.lanes_per_wave, lim.state_mem = 32, A.STATE_GLOBAL
    with pytest.raises(RuntimeError):
        emu.run_batch(LW.ALL["kv_rpc"](), 0, 8, None, lim)
