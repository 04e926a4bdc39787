"""GPU parity: the HIP path through the C-ABI vs the CPU oracle, bit-exact on every result field."""
import numpy as np
import pytest

import oracle
from madsim_amd import _abi as A
from madsim_amd import workload as W
from tests import parity

pytestmark = pytest.mark.gpu


def _cmp(hip, w, seed0, count, config=None, limits=None):
    got, summ = hip.run_batch(w, seed0, count, config, limits)
    want, osumm = oracle.run_batch(w, seed0, count, config, limits)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"seed {seed0 + bad[0]}: gpu {got[bad[0]]} != oracle {want[bad[0]]}"
    assert summ.n_failed == osumm.n_failed
    assert summ.first_failing_seed == osumm.first_failing_seed
    assert summ.total_steps == osumm.total_steps
    return got, summ


# ---- fuzz blocks --------------------------------------------------------------------------------------------------------
# Every test_fuzz_*_gpu runs two blocks of random programs: a FIXED block (the same programs on every run: a regression
# anchor) and a FRESH block whose program seeds derive from MADSIM_FUZZ_SEED — by default the wall clock, so every driver
# run covers programs no earlier run has seen.  A failure message names the generator seed: re-run with
# MADSIM_FUZZ_SEED=<printed value> to reproduce.
def _fuzz_seed():
    import os
    import time
    v = os.environ.get("MADSIM_FUZZ_SEED")
    return int(v, 0) if v else int(time.time())


FUZZ_SEED = _fuzz_seed()


TALLY = parity.Tally()      # per generator: seeds compared, re-run with grown capacities, proven beyond the layout's ceilings


def _fuzz_block(hip, gen, base, n, count, seed_mul, limits, alt_global=False, gen_kw=None, tally=None):
    """Programs gen(Random(base + k)), k < n, `count` seeds each, GPU vs oracle on all 48 result bytes.  EVERY seed is compared:
    a first-pass device capacity verdict (MADSIM_OVERFLOW) is re-run through madsim_hip_run_batch_auto and what comes back is held
    against the oracle like any other seed (tests/parity.py); a seed still OVERFLOW at the largest capacities fails unless the oracle's
    own high-water marks prove it needs more than the layout can hold."""
    import random
    for k in range(n):
        w, cfg, desc = gen(random.Random(base + k), **(gen_kw or {}))
        lim = limits()
        if alt_global and k % 2:
            lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL
        if k % 4 == 3:
            lim.no_trace_hash = 1                      # the reference's plain mode: no determinism-log fingerprint (rand.rs:67)
        parity.gpu_compare(hip, w, k * seed_mul, count, cfg, lim, gen.__name__, TALLY,
                           (f"{gen.__name__}(Random({base + k})) [MADSIM_FUZZ_SEED={FUZZ_SEED}]", desc))
    if tally is not None:                              # (this generator's running totals: seeds, first-pass capacity verdicts, oracle verdicts seen)
        r = TALLY.rows[gen.__name__]
        tally["n"], tally["ovf"], tally["verdicts"] = r[0], r[1], set(TALLY.verdicts)


def _fuzz_two_blocks(hip, gen, fixed_base, n_fixed, n_fresh, salt, **kw):
    _fuzz_block(hip, gen, fixed_base, n_fixed, **kw)
    _fuzz_block(hip, gen, (FUZZ_SEED * 1_000_003 + salt * 7919) & 0x7fffffffffff, n_fresh, **kw)


def _lim_tasks(n):
    def f():
        from tests import fuzz
        lim = fuzz.generous_limits(); lim.max_tasks = n
        return lim
    return f


def test_pingpong_2node_one_seed(hip):
    """BASELINE config 0: 2-node ping-pong, 1 seed, plumbing + bit-exact baseline."""
    _cmp(hip, W.pingpong(2, 64), 0, 1)


def test_pingpong_4node_contiguous(hip):
    _cmp(hip, W.pingpong(4, 64), 0, 4096)


def test_pingpong_4node_65536_sampled(hip):
    """BASELINE config 1: 65 536 seeds on one GPU, cross-checked at 256 sampled seeds k*257 mod 65536."""
    w = W.pingpong(4, 64)
    got, summ = hip.run_batch(w, 0, 65536)
    assert summ.n_failed == 0
    for k in range(256):
        s = (k * 257) % 65536
        want, _ = oracle.run_batch(w, s, 1)
        assert got[s] == want[0], f"seed {s}"


def test_pingpong_loss_first_fail(hip):
    """Fault variant: packet loss => deadlock verdicts; first failing seed must match."""
    cfg = A.Config.default(packet_loss_rate=0.01)
    got, summ = _cmp(hip, W.pingpong(4, 64), 0, 2048, cfg)
    assert summ.n_failed > 0
    assert set(np.unique(got["verdict"])) <= {A.PASS, A.DEADLOCK}


def test_trace_log_bytes(hip):
    """The raw determinism log (rand.rs:64-88) of a seed, byte for byte."""
    w = W.pingpong(2, 4)
    for seed in (0, 1, 12345):
        glog, gres = hip.trace_seed(w, seed)
        olog, ores = oracle.trace_seed(w, seed)
        assert glog == olog
        assert gres.astuple() == ores.astuple()


def test_trace_log_bytes_extended_workloads(hip):
    """The trace build (every op class, general addresses) on the lifecycle / channel / RPC / address workloads: raw log bytes
    and result of a few seeds each, byte for byte — the log the golden fixture holds for the same workloads."""
    for name in ("kill_restart_with_traffic", "exited", "kv_rpc", "channel_wildcard_listener", "ephemeral_clients",
                 "endpoint_bind_ephemeral", "rpc_hooks", "restart_on_panic_matching", "net_ipless_node"):
        w, cfg, lim = LW.ALL[name](), LW.config(name), LW.limits(name)
        for seed in (0, 3):
            glog, gres = hip.trace_seed(w, seed, cfg, lim)
            olog, ores = oracle.trace_seed(w, seed, cfg, lim)
            assert gres.astuple() == ores.astuple(), (name, seed)
            assert glog == olog, (name, seed)


def test_fuzz_random_workloads_gpu(hip):
    """Random actor programs (every verdict, clog/set_loss/close/yield, HBM spill path) through the C-ABI."""
    from tests import fuzz
    t = {"ovf": 0, "n": 0, "verdicts": set()}
    _fuzz_two_blocks(hip, fuzz.random_workload, 5000, 120, 60, 1, count=96, seed_mul=13, limits=fuzz.generous_limits, tally=t)
    assert {A.PASS, A.DEADLOCK, A.PANIC} <= t["verdicts"] and t["ovf"] < 0.02 * t["n"]


@pytest.mark.parametrize("nodes,rounds,count", [(2, 64, 1024), (8, 8, 1024), (16, 4, 512)])
def test_pingpong_topologies(hip, nodes, rounds, count):
    _cmp(hip, W.pingpong(nodes, rounds), 10_000, count)


@pytest.mark.parametrize("cfg", [A.Config.default(buggify=True), A.Config.default(packet_loss_rate=1.0),
                                 A.Config.default(lat_lo_ns=9 * 10**8, lat_hi_ns=21 * 10**8)])
def test_pingpong_configs(hip, cfg):
    _cmp(hip, W.pingpong(4, 16), 3, 1024, cfg)


@pytest.mark.parametrize("lw", [8, 16, 32, 64])
def test_lanes_per_wave_invariant(hip, lw):
    lim = A.Limits(); lim.lanes_per_wave = lw
    _cmp(hip, W.pingpong(4, 8), 0, 2048, None, lim)


def test_heap_spill_path_hbm(hip):
    """Only 2 timer entries per seed in LDS: the rest goes through the coalesced HBM spill region."""
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 30
    _cmp(hip, W.pingpong(8, 8), 0, 4096, None, lim)


def test_overflow_is_reported_not_hidden(hip):
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 0
    got, summ = hip.run_batch(W.pingpong(4, 4), 0, 256, None, lim)
    assert (got["verdict"] == A.OVERFLOW).all() and summ.n_failed == 256


def test_full_size_properties_65536(hip):
    """BASELINE configs[1] at full size through size-independent properties:
    every seed passes, 2 messages and 12 executor steps per round trip, determinism
    (same batch twice), and partition invariance (a batch equals the concatenation of its halves)."""
    w = W.pingpong(4, 64)
    a, sa = hip.run_batch(w, 0, 65536)
    assert sa.n_failed == 0 and sa.first_failing_seed == A.U64_MAX
    assert (a["msg_count"] == 2 * 2 * 64).all()
    assert (a["steps"] >= 2 * 64 * 12).all() and (a["steps"] < 2 * 64 * 12 + 40).all()
    assert len(np.unique(a["trace_hash"])) == 65536
    b, _ = hip.run_batch(w, 0, 65536)
    assert (a == b).all()
    lo, _ = hip.run_batch(w, 0, 30000)
    hi, _ = hip.run_batch(w, 30000, 35536)
    assert (np.concatenate([lo, hi]) == a).all()
    assert sa.total_steps == int(a["steps"].astype(np.int64).sum())
    assert sa.total_clock_ns == int(a["clock_ns"].astype(np.int64).sum())


def test_262144_seeds_sampled(hip):
    """BASELINE configs[2] batch size on one GPU (4-node ping-pong with packet loss as the injected fault)."""
    w = W.pingpong(4, 64)
    cfg = A.Config.default(packet_loss_rate=0.002)
    got, summ = hip.run_batch(w, 0, 262144, cfg)
    fails = np.nonzero(got["verdict"] != A.PASS)[0]
    assert summ.n_failed == len(fails) > 0 and summ.first_failing_seed == fails[0]
    for s in list(fails[:8]) + [(k * 1021) % 262144 for k in range(64)]:
        want, _ = oracle.run_batch(w, int(s), 1, cfg)
        assert got[s] == want[0], f"seed {s}"


def test_device_resident_entry_point_on_torch_stream(hip):
    """madsim_hip_run_batch_device: results stay in HBM (a torch buffer), launched on torch's stream."""
    import torch
    w = W.pingpong(4, 8)
    n = 4096
    buf = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()        # (the fill ran on torch's current stream, the kernel runs on another)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        summ = hip.run_batch_device(w, 77, n, buf.data_ptr(), st.cuda_stream)
    st.synchronize()
    got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
    want, osumm = oracle.run_batch(w, 77, n)
    assert (got == want).all() and summ.total_steps == osumm.total_steps and summ.kernel_ms > 0


def test_builder_run_reports_first_failing_seed(hip, capfd):
    """Builder::run (runtime/builder.rs:121-162): pass -> returns; failure -> the reference's note + raise."""
    from madsim_amd import runtime
    w = W.pingpong(4, 16)
    out = runtime.Builder(seed=5, count=100).run(w)
    assert len(out) == 100
    cfg = A.Config.default(packet_loss_rate=0.05)
    want, osumm = oracle.run_batch(w, 0, 500, cfg)
    with pytest.raises(runtime.SimulationFailure) as ei:
        runtime.Builder(seed=0, count=500, config=cfg).run(w)
    assert ei.value.seed == osumm.first_failing_seed and ei.value.verdict == A.DEADLOCK
    assert f"note: run with `MADSIM_TEST_SEED={osumm.first_failing_seed}` environment variable" in capfd.readouterr().err
    r = runtime.Builder(seed=9, check=True).run(w)          # MADSIM_TEST_CHECK_DETERMINISM
    assert r.verdict == A.PASS


def test_time_limit_and_step_limit(hip):
    wl = W.WorkloadBuilder(); m = wl.main(); m.sleep(secs=10)
    lim = A.Limits(); lim.time_limit_ns = 5 * 10**9
    got, _ = _cmp(hip, wl.build(), 0, 64, None, lim)
    assert (got["verdict"] == A.TIME_LIMIT).all()
    lim = A.Limits(); lim.max_steps = 100
    got, _ = _cmp(hip, W.pingpong(2, 64), 0, 64, None, lim)
    assert (got["verdict"] == A.STEP_LIMIT).all()


def test_reference_property_tests_on_gpu(hip):
    """random_select_from_ready_tasks (task/mod.rs:1018-1041) and deterministic_std_instant
    (time/system_time.rs:140-154) executed by the kernel."""
    wl = W.WorkloadBuilder()
    ts = []
    for i in range(3):
        t = wl.task(0); t.set(0, 5); top = t.label(); t.trace(i * 10, add_reg=0); t.yield_now(); t.djnz(0, top); ts.append(t)
    m = wl.main()
    for t in ts:
        m.spawn(t)
    for t in ts:
        m.join(t)
    got, summ = _cmp(hip, wl.build(), 0, 10)
    assert summ.n_failed == 0 and len(set(got["obs_hash"].tolist())) == 10
    wl = W.WorkloadBuilder(); m = wl.main(); m.mark(); m.sleep(secs=1); m.assert_elapsed("==", secs=1, ns=50)
    got, summ = _cmp(hip, wl.build(), 0, 512)
    assert summ.n_failed == 0


def test_cpp_host_mirror_runs_like_cargo_test(hip):
    """examples/pingpong_test.cpp (C++ Builder mirror): exit 0 on pass, reference-style note on failure."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "pingpong_test")
    if not os.path.exists(exe):
        pytest.skip("example not built (run __graft_entry__.build())")
    hip.shutdown()
    p = subprocess.run([exe], env=dict(os.environ, MADSIM_TEST_SEED="3", MADSIM_TEST_NUM="4096"), capture_output=True, text=True)
    assert p.returncode == 0 and "test ping_pong ... ok (4096 seeds from 3" in p.stdout, p.stderr
    p = subprocess.run([exe], env=dict(os.environ, MADSIM_TEST_SEED="3", MADSIM_TEST_NUM="64", MADSIM_TEST_TIME_LIMIT="1.5"),
                       capture_output=True, text=True)
    assert p.returncode == 101 and "note: run with `MADSIM_TEST_SEED=3` environment variable" in p.stderr
    # typed RPC + server kill/restart through the same mirror (examples/rpc_restart_test.cpp = lifecycle_workloads.rpc_server_restart)
    exe2 = os.path.join(root, "examples", "rpc_restart_test")
    if os.path.exists(exe2):
        p = subprocess.run([exe2], env=dict(os.environ, MADSIM_TEST_SEED="11", MADSIM_TEST_NUM="2048"), capture_output=True, text=True)
        assert p.returncode == 0 and "test rpc_survives_restart ... ok (2048 seeds from 11)" in p.stdout, p.stderr
    # reliable connections from ephemeral ports to a 0.0.0.0 listener (examples/kv_client_test.cpp): under 5 % loss some seeds fail
    exe3 = os.path.join(root, "examples", "kv_client_test")
    if os.path.exists(exe3):
        p = subprocess.run([exe3], env=dict(os.environ, MADSIM_TEST_SEED="5", MADSIM_TEST_NUM="2048"), capture_output=True, text=True)
        assert p.returncode == 0 and "test kv_requests ... ok (2048 seeds from 5)" in p.stdout, p.stderr
    # IPVS round robin over connect1 + a node restarting on substring panic patterns (examples/ipvs_test.cpp)
    exe4 = os.path.join(root, "examples", "ipvs_test")
    if os.path.exists(exe4):
        p = subprocess.run([exe4], env=dict(os.environ, MADSIM_TEST_SEED="7", MADSIM_TEST_NUM="2048"), capture_output=True, text=True)
        assert p.returncode == 0 and "test ipvs_load_balance ... ok (2048 seeds from 7)" in p.stdout, p.stderr
    hip.init(0)


from tests import lifecycle_workloads as LW  # noqa: E402


@pytest.mark.parametrize("name", sorted(LW.ALL))
def test_lifecycle_reference_tests_gpu(hip, name):
    """The reference's node-lifecycle unit tests (task/mod.rs:859-1182) executed by the kernel, 1024 seeds each."""
    got, _ = _cmp(hip, LW.ALL[name](), 0, 1024, LW.config(name), LW.limits(name))
    assert (got["verdict"] == (A.PANIC if name in LW.EXPECT_PANIC else A.PASS)).all()


def test_fuzz_lifecycle_workloads_gpu(hip):
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_lifecycle_workload, 7000, 150, 75, 2, count=64, seed_mul=17, limits=_lim_tasks(24))


def test_fuzz_guard_workloads_gpu(hip):
    """Random lifecycle programs whose task bodies own guards that spawn in Drop (MADSIM_PROG_DROP_SPAWN, task/mod.rs:1184-1253);
    odd rounds with the per-seed state in the global-memory block."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_guard_workload, 7600, 150, 75, 3, count=64, seed_mul=17, limits=_lim_tasks(24), alt_global=True)


def test_fuzz_supervisor_workloads_gpu(hip):
    """Supervisor calls from every task: task::spawn after killing / restarting the own node, JoinHandles awaited across a
    respawn of their program, init tasks spawned by hand; odd rounds with the per-seed state in the global-memory block."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_supervisor_workload, 7800, 200, 100, 4, count=64, seed_mul=17, limits=_lim_tasks(48), alt_global=True)


def test_fuzz_latency_workloads_gpu(hip):
    """NetSim::update_config of send_latency (MS_OP_SET_LATENCY) from the supervisor and the senders, both
    UniformDuration paths, under timeouts / clogs / loss; odd rounds with the per-seed state in the global-memory block."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_latency_workload, 61000, 200, 100, 17, count=64, seed_mul=5, limits=fuzz.mailbox_limits, alt_global=True)


def test_fuzz_mixed_workloads_gpu(hip):
    """Everything from everywhere — supervisor calls, datagrams, channel and RPC exchanges, service tasks (echo, RPC handler,
    accept loop) — from every task: the deepest generator, LDS-resident (even rounds) and global-state (odd rounds) builds."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_mixed_workload, 9900, 200, 100, 5, count=64, seed_mul=3, limits=fuzz.mixed_limits, alt_global=True)


def test_config2_election_loop_262144_seeds(hip):
    """BASELINE configs[2]: 5-node election loop with NetSim partition injection, 262 144 seeds on one GPU
    (timeout() duplicate timers push most of the timer heap into the HBM spill region)."""
    w, lim = W.raft_election(), W.raft_election_limits()
    # capacities sized to the common case; the rare seed that outgrows them (dead registrations pile up: the
    # reference's Vec is unbounded) comes back MADSIM_OVERFLOW and is re-run with doubled capacities
    first, _ = hip.run_batch(w, 0, 262144, None, lim)
    n_over = int((first["verdict"] == A.OVERFLOW).sum())
    assert n_over < 262144 // 100
    got, summ = hip.run_batch_auto(w, 0, 262144, None, lim)
    assert summ.n_failed == 0 and (got["verdict"] == A.PASS).all()
    assert len(np.unique(got["trace_hash"])) == 262144
    over = np.nonzero(first["verdict"] == A.OVERFLOW)[0][:16]
    for s in [(k * 4093) % 262144 for k in range(96)] + [int(i) for i in over]:
        want, _ = oracle.run_batch(w, s, 1)
        assert got[s] == want[0], f"seed {s}"
    big = hip.grow_limits(lim)
    _cmp(hip, w, 1_000_000, 2048, A.Config.default(packet_loss_rate=0.05), big)


def test_config3_kv_rpc_131072_seeds(hip):
    """BASELINE configs[3] per-GPU share (1 048 576 seeds / 8 GPUs): etcd-style KV ops over the reliable channel."""
    w, lim = W.kv_rpc(), W.kv_rpc_limits()
    got, summ = hip.run_batch(w, 0, 131072, None, lim)
    assert summ.n_failed == 0
    for s in [(k * 2039) % 131072 for k in range(96)]:
        want, _ = oracle.run_batch(w, s, 1, None, lim)
        assert got[s] == want[0], f"seed {s}"
    _cmp(hip, w, 5_000_000, 2048, A.Config.default(packet_loss_rate=0.02), lim)
    # service.timeout() firing (timeout_rate 0.2): 5-15 s sleeps and error responses
    _cmp(hip, w, 6_000_000, 2048, A.Config.default(loss_table=(0.0, 0.2, 1.0)), lim)


def test_config4_streaming_topology_524288_seeds(hip):
    """BASELINE configs[4] per-GPU share (4 194 304 seeds / 8 GPUs): 16-node streaming topology (etcd-style meta service,
    typed-RPC brokers, 12 compute nodes, broker clog + restart), event heap mostly in the HBM spill region."""
    w, lim = W.streaming_topology(), W.streaming_topology_limits()
    g = hip.geometry(w, lim)
    assert g.heap_spill_slots > 4 * g.heap_lds_slots and (g.variant >> 8) & 0x80       # mostly spilled; 8-byte entries since round 6
    n = 524288
    got, summ = hip.run_batch_auto(w, 0, n, None, lim)
    assert summ.n_failed == 0 and (got["verdict"] == A.PASS).all()
    assert len(np.unique(got["trace_hash"])) == n                      # every seed took its own path
    assert (got["msg_count"] >= 12 * 2).all()                          # at least the 12 registrations with meta
    for s in [(k * 8191) % n for k in range(64)]:
        want, _ = oracle.run_batch(w, s, 1)
        assert got[s] == want[0], f"seed {s}"
    _cmp(hip, w, 7_000_000, 1024, A.Config.default(packet_loss_rate=0.03), hip.grow_limits(lim))


def test_async_entry_point_device_summary(hip):
    """madsim_hip_run_batch_async: no host copies; the 4-word report lands in HBM in all-reduce-ready form."""
    import torch
    from madsim_amd import dist as mdist
    w = W.pingpong(4, 16)
    cfg = A.Config.default(packet_loss_rate=0.02)
    n = 8192
    buf = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    rep = torch.zeros(4, dtype=torch.int64, device="cuda")
    hip.run_batch_async(w, 1000, n, buf.data_ptr(), rep.data_ptr(), torch.cuda.current_stream().cuda_stream, cfg, None, timing_slot=3)
    torch.cuda.synchronize()
    want, osumm = oracle.run_batch(w, 1000, n, cfg)
    got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
    assert (got == want).all()
    r = rep.cpu()
    assert (mdist.decode_first_fail(r[0]), int(r[1]), int(r[2]), int(r[3])) == \
        (osumm.first_failing_seed, osumm.n_failed, osumm.total_steps, osumm.total_clock_ns)
    assert osumm.n_failed > 0 and hip.timing_ms(3) > 0
    hip.run_batch_async(w, 0, n, buf.data_ptr(), rep.data_ptr(), torch.cuda.current_stream().cuda_stream)   # no failure
    torch.cuda.synchronize()
    assert mdist.decode_first_fail(rep.cpu()[0]) == A.U64_MAX


def test_capacity_verdicts_are_retried_with_larger_limits(hip):
    """Builder.run never surfaces a device-capacity verdict as the answer: overflowed seeds are re-run with
    doubled capacities (the reference's containers are unbounded)."""
    from madsim_amd import runtime
    w = W.raft_election()
    tight = A.Limits(); tight.heap_lds_slots, tight.heap_spill_slots = 8, 8; tight.mbox_regs, tight.mbox_msgs = 4, 4
    first, _ = hip.run_batch(w, 0, 256, None, tight)
    assert (first["verdict"] == A.OVERFLOW).any()
    got, summ = runtime.run_batch_auto(w, 0, 256, None, tight, max_rounds=6)
    want, osumm = oracle.run_batch(w, 0, 256)
    assert (got == want).all() and summ.n_failed == osumm.n_failed == 0


def test_concurrent_streams_do_not_share_scratch(hip):
    """Launches in flight on different HIP streams: each has its own heap-spill region and keeps its own workload
    tables (two workloads alternate), so overlapping batches stay bit-exact."""
    import torch
    wa, la = W.timer_storm(), W.timer_storm_limits(4)          # most of the 24-entry heap spills to HBM
    wb = W.pingpong(4, 16)
    n = 16384
    streams = [torch.cuda.Stream() for _ in range(3)]
    jobs = []
    bufs = [torch.zeros(n * 48, dtype=torch.uint8, device="cuda") for _ in range(6)]
    reps = [torch.zeros(4, dtype=torch.int64, device="cuda") for _ in range(6)]
    torch.cuda.synchronize()        # the fills run on torch's current stream: they must not land after a kernel on another stream
    for k in range(6):
        w, lim = (wa, la) if k % 2 == 0 else (wb, None)
        buf, rep = bufs[k], reps[k]
        st = streams[k % 3]
        with torch.cuda.stream(st):
            hip.run_batch_async(w, 7000 * k, n, buf.data_ptr(), rep.data_ptr(), st.cuda_stream, None, lim)
        jobs.append((w, lim, 7000 * k, buf))
    torch.cuda.synchronize()
    for w, lim, seed0, buf in jobs:
        got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
        idx = np.arange(0, n, 37)
        want = np.concatenate([oracle.run_batch(w, seed0 + int(i), 1, None, lim)[0] for i in idx])
        assert (got[idx] == want).all()
        assert (got["verdict"] == A.PASS).all()


def test_register_ready_queue_with_heap_spill(hip):
    """Variant<SPILL, RQ>: 2 LDS heap slots + HBM spill under the register-resident ready queue."""
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots, lim.mbox_regs, lim.mbox_msgs = 2, 6, 1, A.LIMIT_NONE
    w = W.pingpong(4, 32)
    assert hip.geometry(w, lim).variant & 0xfff == 5          # heap spill + register ready queue, base ops
    _cmp(hip, w, 424242, 4096, None, lim)


def test_fuzz_rpc_workloads_gpu(hip):
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_rpc_workload, 33000, 120, 60, 6, count=96, seed_mul=19, limits=_lim_tasks(24))


def test_rpc_echo_65536_seeds(hip):
    """Typed RPC at batch size: 2 callers x 4 calls against one handler task, sampled parity + size-independent
    properties (every call costs one request and one response: 16 messages per seed without loss)."""
    from tests import lifecycle_workloads as LW
    w = LW.rpc_echo()
    n = 65536
    got, summ = hip.run_batch(w, 9_000_000, n)
    assert summ.n_failed == 0 and (got["msg_count"] == 16).all()
    idx = np.arange(0, n, 257)
    want = np.concatenate([oracle.run_batch(w, 9_000_000 + int(i), 1)[0] for i in idx])
    assert (got[idx] == want).all()


@pytest.mark.parametrize("count", [0, 1, 63, 65, 255, 4097])
def test_ragged_and_empty_batches(hip, count):
    """Batch sizes that do not fill a wave / a workgroup, and the empty batch."""
    w = W.pingpong(4, 8)
    got, summ = hip.run_batch(w, 77, count)
    want, osumm = oracle.run_batch(w, 77, count)
    assert len(got) == count and (got == want).all()
    assert (summ.n_failed, summ.first_failing_seed, summ.total_steps) == (osumm.n_failed, osumm.first_failing_seed, osumm.total_steps)


def test_largest_topology(hip):
    """The largest ping-pong the device tables admit: 30 nodes (31 with the supervisor), 31 programs, 30 sockets."""
    _cmp(hip, W.pingpong(30, 2), 5, 256)


def test_seed_range_extremes(hip):
    """Seeds near 0 and near 2^64 (seed arithmetic is u64, the first-fail key flips the sign bit)."""
    w = W.pingpong(2, 4)
    cfg = A.Config.default(packet_loss_rate=0.3)
    top = (1 << 64) - 600
    got, summ = hip.run_batch(w, top, 512, cfg)
    want, osumm = oracle.run_batch(w, top, 512, cfg)
    assert (got == want).all() and osumm.n_failed > 0
    assert summ.first_failing_seed == osumm.first_failing_seed and summ.n_failed == osumm.n_failed


@pytest.mark.parametrize("n_streams,state_mem", [(1, 0), (2, 0), (5, 0), (2, 1)])
def test_bench_configuration_pingpong_is_oracle_checked(hip, n_streams, state_mem):
    """The exact configuration bench.py times (workload.bench_case: heap 4 LDS / 0 spill, mbox_regs 1, mbox_msgs NONE; the
    compact layout by AUTO: Variant<false,false,6,COMPACT,true>, 152 B per seed, four waves per SIMD — and the plain layout
    it replaced, state_mem = LDS), 65 536 seeds per launch, 1, 2 and 5 launches in flight through the async entry point:
    256 sampled seeds k*257 mod 65 536 and 4 096 contiguous seeds of every launch against the oracle."""
    import torch
    w, lim, _ = W.bench_case("pingpong")
    lim.state_mem = state_mem
    g = hip.geometry(w, lim)
    if state_mem == 0:
        assert g.variant & 0xff == 4 and (g.variant >> 14) & 1 and hip.variant_name(g) == "sim_kernel<Variant<false, false, 6, 64, true, false>>" and g.lds_bytes_per_seed == 152
    else:
        assert g.variant & 0xfff == 4 and hip.variant_name(g) == "sim_kernel<Variant<false, false, 6, 0, true, false>>" and g.lds_bytes_per_seed == 200
    assert g.lanes_per_wave == 64 and g.heap_lds_slots == 4 and g.heap_spill_slots == 0
    n = W.BENCH_SEEDS_PER_GPU
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    bufs = [torch.zeros(n * 48, dtype=torch.uint8, device="cuda") for _ in range(2 * n_streams)]
    reps = [torch.zeros(4, dtype=torch.int64, device="cuda") for _ in range(2 * n_streams)]
    torch.cuda.synchronize()        # (the fills ran on torch's current stream, the kernels run on others)
    for k in range(2 * n_streams):                     # two rounds per stream, all queued before any finishes
        st = streams[k % n_streams]
        with torch.cuda.stream(st):
            hip.run_batch_async(w, 5_000_000 + k * n, n, bufs[k].data_ptr(), reps[k].data_ptr(), st.cuda_stream, None, lim)
    torch.cuda.synchronize()
    for k in range(2 * n_streams):
        got = np.frombuffer(bufs[k].cpu().numpy().tobytes(), dtype=A.RESULT_DTYPE)
        base = 5_000_000 + k * n
        assert (got["verdict"] == A.PASS).all()
        idx = sorted({(j * 257) % n for j in range(256)})
        want = np.concatenate([oracle.run_batch(w, base + i, 1, None, lim)[0] for i in idx])
        assert (got[idx] == want).all()
        lo = (k * 7919) % (n - 4096)
        want, osm = oracle.run_batch(w, base + lo, 4096, None, lim)
        assert (got[lo:lo + 4096] == want).all()
        rep = reps[k].cpu().tolist()
        assert rep[1] == 0 and rep[2] == int(got["steps"].astype(np.int64).sum())


@pytest.mark.parametrize("name", ["raft", "kv", "timers", "topo"])
def test_bench_configuration_extras_are_oracle_checked(hip, name):
    """bench.py --workload raft / kv / timers / topo: the same (workload, limits) objects, 16 384 seeds, 128 sampled.
    A seed that outgrew a device capacity in the plain call (bench.py counts those as failed) is re-run with grown capacities and
    compared like the others; no seed stays MADSIM_OVERFLOW."""
    w, lim, _ = W.bench_case(name)
    n = 16384
    got, n_ovf = parity.run_resolved(hip, w, 31_000_000, n, None, lim)
    idx = np.arange(0, n, 128)
    want = np.concatenate([oracle.run_batch(w, 31_000_000 + int(i), 1, None, lim)[0] for i in idx])
    assert (got[idx] == want).all() and n_ovf < 0.005 * n, n_ovf


def test_contexts_multi_gpu_entry_matches_single_context(hip):
    """madsim_hip_run_batch_multi: the seed range sharded over several contexts from ONE host thread (on a 1-GPU box all
    contexts sit on GPU 0 — the same code path as one per GPU), bit-identical to the single-context run and the oracle;
    ragged shard sizes, packet loss (first failing seed), fewer seeds than contexts."""
    w = W.pingpong(4, 16)
    cfg = A.Config.default(packet_loss_rate=0.01)
    ctxs = [hip.Context(0) for _ in range(3)]
    try:
        for seed0, count in ((123, 10001), (5, 2), (9, 0)):
            want, osm = oracle.run_batch(w, seed0, count, cfg)
            one, s1 = ctxs[0].run_batch(w, seed0, count, cfg, auto_rounds=5)
            got, sn = hip.run_batch_multi(ctxs, w, seed0, count, cfg)
            assert (got == want).all() and (one == want).all()
            for s in (s1, sn):
                assert (s.first_failing_seed, s.n_failed, s.total_steps, s.total_clock_ns) == \
                       (osm.first_failing_seed, osm.n_failed, osm.total_steps, osm.total_clock_ns)
        # contexts are independent: interleaved use, different workloads, and the default context still works
        a, _ = ctxs[1].run_batch(W.pingpong(2, 8), 0, 512)
        b, _ = ctxs[2].run_batch(W.pingpong(8, 4), 0, 512)
        c, _ = hip.run_batch(W.pingpong(2, 8), 0, 512)
        assert (a == c).all() and (b == oracle.run_batch(W.pingpong(8, 4), 0, 512)[0]).all()
    finally:
        for c in ctxs:
            c.close()


def test_runner_verdicts_are_rerun_compacted(hip):
    """run_batch_auto: seeds that outgrow a device capacity (OVERFLOW) or reach the step cap (STEP_LIMIT) are gathered into
    one compacted re-launch per round (seed-list indirection) and end with the oracle's answer — also when they are
    sparse and scattered, which used to cost one launch per run of overflowed seeds."""
    w, lim, _ = W.bench_case("raft")
    lim.mbox_regs = 24                   # ~5 % of the seeds leave more than 24 dead registrations on a socket (oracle high-water marks)
    n = 4096
    first, _ = hip.run_batch(w, 0, n, None, lim)
    n_ovf = int((first["verdict"] == A.OVERFLOW).sum())
    assert 0 < n_ovf < n // 4
    got, summ = hip.run_batch_auto(w, 0, n, None, lim)
    want, osm = oracle.run_batch(w, 0, n)
    assert (got == want).all() and (summ.n_failed, summ.first_failing_seed) == (osm.n_failed, osm.first_failing_seed)
    w = W.pingpong(4, 16)
    # the step cap is a runner limit too (the reference has none): 100 steps -> re-run with 1 600 -> 25 600 ...
    lim = A.Limits(); lim.max_steps = 100
    got, summ = hip.run_batch_auto(w, 0, 4096, None, lim)
    want, _ = oracle.run_batch(w, 0, 4096)
    assert (got == want).all() and summ.n_failed == 0
    # and Builder.run never reports a runner verdict as a test failure
    from madsim_amd import runtime
    wl = W.WorkloadBuilder(); m = wl.main(); m.set(0, 40000); top = m.label(); m.sleep(ms=1); m.djnz(0, top)
    out = runtime.Builder(seed=1, count=8).run(wl.build())          # 80 000 steps per seed: passes, like in madsim
    assert (out["verdict"] == A.PASS).all() and (out["steps"] > 80000).all()


@pytest.mark.parametrize("sched", [A.SCHED_STATIC, A.SCHED_QUEUE])
def test_work_distribution_is_result_invariant(hip, sched):
    """More seeds than resident lanes in ONE launch: static striding and the atomic work queue give the oracle's results
    (which lane runs a seed can never matter). Packet loss makes seeds end at very different times."""
    w, lim, _ = W.bench_case("pingpong")
    lim.sched = sched
    cfg = A.Config.default(packet_loss_rate=0.01)
    n = 400_000                                            # > 2 waves per SIMD x 64 lanes x 1 024 SIMDs
    got, summ = hip.run_batch(w, 1 << 33, n, cfg, lim)
    idx = np.concatenate([np.arange(0, n, 997), np.arange(n - 300, n)])
    want = np.concatenate([oracle.run_batch(w, (1 << 33) + int(i), 1, cfg, lim)[0] for i in idx])
    assert (got[idx] == want).all()
    assert summ.n_failed == int((got["verdict"] != A.PASS).sum()) > 0


def test_plain_c_client_drives_two_contexts(hip):
    """examples/multi_ctx_test.c (plain C, include/madsim_hip.h only): two / five contexts vs one, incl. overflow re-runs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "multi_ctx_test")
    if not os.path.exists(exe):
        pytest.skip("example not built (run __graft_entry__.build())")
    for n_ctx, count in (("2", "10000"), ("5", "33333")):
        p = subprocess.run([exe, n_ctx, count], capture_output=True, text=True)
        assert p.returncode == 0 and "identical to the single-context run" in p.stdout, p.stdout + p.stderr


def test_ref_twin_workloads_gpu(hip):
    """tools/ref_twin: the tables whose Rust originals run on real madsim — the kernel must produce the oracle's
    fingerprint (elapsed, msg_count, trailing draw folded into obs_hash) on each of them."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ref_twin"))
    import twin_workloads as T
    for name in sorted(T.ALL):
        _cmp(hip, T.ALL[name](), 0, 512, T.config(name), T.limits(name))
    _cmp(hip, T.ALL["pingpong4"](), 0, 512, A.Config.default(packet_loss_rate=0.01))


def test_fuzz_address_resolution_gpu(hip):
    """Random datagram programs over mixed address kinds (network.rs:206-313: wildcard fallback, loopback, IP-less nodes)."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_addr_workload, 76000, 150, 75, 7, count=96, seed_mul=29, limits=fuzz.generous_limits, alt_global=True)


def test_fuzz_ephemeral_ports_gpu(hip):
    """Random programs binding port 0 (network.rs:224-236): literal port hand-out in the oracle, candidate entries on the GPU."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_ephemeral_workload, 77000, 150, 75, 8, count=96, seed_mul=31, limits=fuzz.generous_limits, alt_global=True)


def test_fuzz_ipvs_gpu(hip):
    """Random programs over IP Virtual Server rewriting (net/ipvs.rs; NetSim::send / connect1, net/mod.rs:312-317,345-350)."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_ipvs_workload, 79000, 150, 75, 13, count=96, seed_mul=41, limits=_lim_tasks(24), alt_global=True)


def test_fuzz_ipvs_runtime_gpu(hip):
    """Random programs whose operator tasks change IPVS services at run time (MS_OP_IPVS: add_service / del_service /
    add_server / del_server, net/ipvs.rs:50-85), both state layouts."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_ipvs_runtime_workload, 79500, 150, 75, 14, count=96, seed_mul=43, limits=_lim_tasks(24), alt_global=True)


def test_fuzz_channel_guards_gpu(hip):
    """Random reliable-channel programs about who keeps an address bound (Arc<BindGuard> clones in Sender / Receiver)."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_channel_workload, 78000, 150, 75, 9, count=96, seed_mul=37, limits=_lim_tasks(24), alt_global=True)


def test_fuzz_rpc_hooks_gpu(hip):
    """Random typed-RPC programs with NetSim::hook_rpc_req / hook_rpc_rsp (net/mod.rs:240-284), LDS and global state."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_rpc_workload, 64000, 120, 60, 10, count=96, seed_mul=19, limits=_lim_tasks(24), alt_global=True, gen_kw={"hooks": True})


def _global_limits(lim=None):
    g = A.Limits()
    if lim is not None:
        for f, _ in A.Limits._fields_:
            setattr(g, f, getattr(lim, f))
    g.lanes_per_wave, g.state_mem = 0, A.STATE_GLOBAL
    return g


@pytest.mark.parametrize("name", sorted(LW.ALL))
def test_global_state_lifecycle_reference_tests_gpu(hip, name):
    """The reference's lifecycle / channel / RPC tests with the task table and planes in the per-lane global-memory block
    (Variant::G) instead of LDS: same results, bit for bit."""
    lim = _global_limits(LW.limits(name))
    g = hip.geometry(LW.ALL[name](), lim)
    assert bool(g.variant & 16) == bool(g.variant & 2)        # every extended-op workload takes the global-state build when asked
    got, _ = _cmp(hip, LW.ALL[name](), 0, 1024, LW.config(name), lim)
    assert (got["verdict"] == (A.PANIC if name in LW.EXPECT_PANIC else A.PASS)).all()


def test_global_state_fuzz_gpu(hip):
    from tests import fuzz

    def lim():
        g = _global_limits(fuzz.generous_limits()); g.max_tasks = 24
        return g
    for gen, base, salt in ((fuzz.random_rpc_workload, 52000, 11), (fuzz.random_lifecycle_workload, 52500, 12)):
        _fuzz_two_blocks(hip, gen, base, 80, 40, salt, count=128, seed_mul=23, limits=lim)


@pytest.mark.parametrize("name", ["raft", "kv", "topo"])
def test_global_and_lds_state_agree_at_batch_size(hip, name):
    """configs[2]/[3]/[4]-shaped workloads, 32 768 seeds: the global-state build (what bench.py runs) and the LDS-resident
    build give identical results; 96 sampled seeds against the oracle."""
    w, lim, _ = W.bench_case(name)
    n = 32768
    assert hip.geometry(w, lim).variant & 16
    a, na = parity.run_resolved(hip, w, 77_000_000, n, None, lim)
    lim.state_mem = A.STATE_LDS
    assert hip.geometry(w, lim).variant & 16 == 0
    b, nb = parity.run_resolved(hip, w, 77_000_000, n, None, lim)
    assert (a == b).all() and na + nb < 0.01 * n               # (capacity verdicts of either first pass: re-run, then every seed compared)
    idx = np.arange(0, n, 343)
    want = np.concatenate([oracle.run_batch(w, 77_000_000 + int(i), 1, None, lim)[0] for i in idx])
    assert (a[idx] == want).all()


def test_bench_two_ranks_share_the_gpu_over_gloo(hip):
    """The N > 1 path of bench.py end to end on a 1-GPU box: `--gpus 2` re-executes itself as two ranks
    (torch.distributed.run), both on this GPU, reports reduced over gloo (the functional-test hook; a real node runs one
    rank per GPU over RCCL).  The line must say n_gpus 2, cover both ranks' seed blocks and carry oracle-verified seeds."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip.shutdown()
    env = dict(os.environ, MADSIM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-measure-traffic"], env=env, capture_output=True, text=True, timeout=600)
    hip.init(0)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    assert line["config"]["seeds_per_step"] == 2 * 65536 and line["verified_seeds"] >= 2 * 256
    assert line["extra"]["failed_seeds"] == 0 and line["extra"]["seeds_per_sec"] > 0
    # ... and the seed SEARCH through the same ranks (runtime.run_campaign_over_ranks): pipelined chunks per rank, one all-gather per round
    ff = line["extra"]["first_fail_over_ranks"]
    assert ff["world"] == 2 and ff["failed"] == 0 and ff["rounds"] == 3 and ff["seeds_searched"] == 3 * 2 * ff["round_batches_per_rank"] * 65536
    assert line["extra"]["first_fail_seeds_per_hour"] == ff["seeds_per_hour"] > 0


def test_campaign_keeps_batches_in_flight_and_reports_like_the_oracle(hip):
    """madsim_hip_run_campaign: batches on the library's own streams, reports folded in order.  Without STOP the report is the
    oracle's over the whole range; with STOP_AT_FAILURE it is the oracle's first failing seed, found after reading exactly the
    batches up to it, at most in_flight - 1 launched beyond; runner verdicts are counted apart and never stop a campaign."""
    w = W.pingpong(4, 16)
    cfg = A.Config.default(packet_loss_rate=0.002)
    total, batch = 40_000, 4096                                     # ragged last batch
    rep = hip.run_campaign(w, 5_000_000, total, batch, 3, False, cfg)
    want, osum = oracle.run_batch(w, 5_000_000, total, cfg)
    assert (rep.seeds_run, rep.batches_run, rep.batches_launched) == (total, 10, 10)
    assert (rep.n_failed, rep.first_failing_seed, rep.total_steps, rep.n_runner) == (osum.n_failed, osum.first_failing_seed, osum.total_steps, 0)
    assert rep.n_failed > 0 and rep.total_clock_ns == int(want["clock_ns"].sum())
    # rare failures: stop at the first batch that has one
    cfg = A.Config.default(packet_loss_rate=0.000002)
    rep = hip.run_campaign(w, 9_000_000, 64 * batch, batch, 3, True, cfg)
    assert rep.first_failing_seed != (1 << 64) - 1
    off = rep.first_failing_seed - 9_000_000
    j = off // batch
    assert rep.batches_run == j + 1 and rep.seeds_run == (j + 1) * batch and j + 1 <= rep.batches_launched <= j + 3
    want, _ = oracle.run_batch(w, 9_000_000, off + 1, cfg)
    assert (want["verdict"][:-1] == A.PASS).all() and want["verdict"][-1] != A.PASS
    # a capacity nobody fits: runner verdicts only
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 0
    rep = hip.run_campaign(w, 0, 3 * batch, batch, 2, True, None, lim)
    assert (rep.n_runner, rep.n_failed, rep.batches_run) == (3 * batch, 0, 3) and rep.first_failing_seed == (1 << 64) - 1


def test_run_batch_multi_from_two_threads_with_permuted_context_lists(hip):
    """madsim_hip_run_batch_multi takes its context locks in address order: two host threads handing it the same contexts in
    opposite orders must both finish (no lock-order deadlock) with the single-context answer; an error after the first
    launch (a workload the second validation refuses is impossible to stage, so: a capacity the device cannot fit) comes back
    as an error code with nothing left in flight — the next call on the same contexts works."""
    import threading
    w = W.pingpong(4, 8)
    ref, rsum = hip.run_batch(w, 1234, 6000)
    with hip.Context(0) as c0, hip.Context(0) as c1:
        outs, errs = {}, []

        def work(tag, ctxs):
            try:
                for _ in range(6):
                    outs[tag] = hip.run_batch_multi(ctxs, w, 1234, 6000)
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        ta = threading.Thread(target=work, args=("a", [c0, c1])); tb = threading.Thread(target=work, args=("b", [c1, c0]))
        ta.start(); tb.start(); ta.join(120); tb.join(120)
        assert not ta.is_alive() and not tb.is_alive(), "deadlock: the two calls never returned"
        assert not errs, errs
        for tag in ("a", "b"):
            got, summ = outs[tag]
            assert (got == ref).all() and summ.first_failing_seed == rsum.first_failing_seed and summ.total_steps == rsum.total_steps
        bad = A.Limits(); bad.heap_lds_slots = 100000                    # per-seed LDS far beyond a CU: make_geometry refuses it
        with pytest.raises(hip.MadsimHipError):
            hip.run_batch_multi([c0, c1], w, 1234, 6000, None, bad)
        got, _ = hip.run_batch_multi([c0, c1], w, 1234, 6000)            # the contexts are intact
        assert (got == ref).all()


@pytest.mark.parametrize("case", ["pingpong", "spill", "lanes16", "raft", "kv", "topo", "timers"])
def test_no_trace_hash_drops_only_the_fingerprint(hip, case):
    """madsim_limits_t.no_trace_hash = the reference's plain run (rand.rs:67: log and check both None): trace_hash comes back 0
    and every other field is what the logging run reports, GPU == oracle in both modes — on the base-op build compiled
    without the fold (geometry reports it) and on the builds that test the flag at run time; a trace request logs anyway."""
    lim = None
    if case == "pingpong":
        w, lim, _ = W.bench_case("pingpong", 4, 64, 4)
    elif case == "spill":
        w = W.pingpong(8, 4); lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 2, 30
    elif case == "lanes16":
        w = W.pingpong(4, 8); lim = A.Limits(); lim.lanes_per_wave = 16
    else:
        w, lim, _ = W.bench_case(case, 4, 64, 4)
    n = 256 if case in ("raft", "topo") else 4096
    with_log, _ = _cmp(hip, w, 11, n, None, lim)
    lim = lim or A.Limits()
    lim.no_trace_hash = 1
    without, _ = _cmp(hip, w, 11, n, None, lim)
    assert (without["trace_hash"] == 0).all() and (with_log["trace_hash"] != 0).all()
    for f in ("verdict", "steps", "clock_ns", "msg_count", "rng_calls", "obs_hash"):
        assert (without[f] == with_log[f]).all(), f
    g = hip.geometry(w, lim)
    assert bool((g.variant >> 13) & 1) == (case == "pingpong"), hip.variant_name(g)
    glog, gres = hip.trace_seed(w, 11, None, lim)
    olog, ores = oracle.trace_seed(w, 11, None, lim)
    assert bytes(glog) == bytes(olog) and len(olog) > 0 and gres.astuple() == ores.astuple()
    assert ores.trace_hash == with_log["trace_hash"][0]


def _compact(lim):
    lim.state_mem = A.STATE_COMPACT
    return lim


@pytest.mark.parametrize("nodes,rounds,loss,n", [(4, 64, 0.0, 65536), (2, 64, 0.0, 4096), (6, 9, 0.0, 4096), (4, 16, 0.05, 4096), (4, 16, 1.0, 1024)])
def test_compact_layout_pingpong(hip, nodes, rounds, loss, n):
    """MADSIM_STATE_COMPACT (8-byte heap entries on the low deadline word, heap root in registers, main task in global memory):
    the oracle's answers on every field, with and without the determinism-log fold; 4-node / 65 536 seeds = the bench batch."""
    w, lim, _ = W.bench_case("pingpong", nodes, rounds, 4)
    lim.heap_lds_slots = max(4, nodes)
    cfg = A.Config.default(packet_loss_rate=loss) if loss else None
    _compact(lim)
    g = hip.geometry(w, lim)
    assert g.lds_bytes_per_seed == (lim.heap_lds_slots - 1) * 8 + nodes * 24 + nodes * 8 and (g.variant >> 14) & 1
    if n > 8192:                                    # the whole batch on the GPU, 512 sampled seeds on the oracle
        got, _ = hip.run_batch(w, 1 << 33, n, cfg, lim)
        for j in range(512):
            i = (j * 257) % n
            want, _ = oracle.run_batch(w, (1 << 33) + i, 1, cfg, lim)
            assert got[i] == want[0], (i, got[i], want[0])
    else:
        _cmp(hip, w, 5, n, cfg, lim)
    lim.no_trace_hash = 1
    _cmp(hip, w, 5, min(n, 4096), cfg, lim)


def test_compact_layout_selection_and_horizon(hip):
    w, lim, _ = W.bench_case("pingpong", 4, 64, 4)
    g = hip.geometry(w, lim)
    assert g.lds_bytes_per_seed == 152 and g.blocks_per_cu * g.block_threads // 64 == 16     # auto: four waves per SIMD
    lim.state_mem = A.STATE_LDS
    g = hip.geometry(w, lim)
    assert g.lds_bytes_per_seed == 200 and g.blocks_per_cu * g.block_threads // 64 == 12     # the plain layout stays selectable
    wl = W.WorkloadBuilder()
    n1 = wl.create_node()
    t = wl.task(n1); t.sleep(secs=3)
    m = wl.main(); m.spawn(t); m.join(t)
    far = wl.build()
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 4, 0
    _cmp(hip, far, 0, 256, None, lim)                    # auto: outside the 2^31 ns horizon, the plain layout runs it
    with pytest.raises(RuntimeError):
        hip.run_batch(far, 0, 256, None, _compact(lim))
    w, lim, _ = W.bench_case("pingpong", 4, 8, 4)
    with pytest.raises(RuntimeError):
        hip.run_batch(w, 0, 256, A.Config.default(buggify=True), _compact(lim))
    lim.state_mem = A.STATE_AUTO
    _cmp(hip, w, 0, 1024, A.Config.default(buggify=True), lim)


def test_compact_layout_fuzz_gpu(hip):
    """Random base-op programs small enough for the compact layout, compact vs oracle; fixed block + fresh block."""
    import random
    from tests import fuzz
    ran = 0
    for base, n in ((31000, 120), ((FUZZ_SEED * 1_000_003 + 77 * 7919) & 0x7fffffffffff, 120)):
        for k in range(n):
            w, cfg, desc = fuzz.random_workload(random.Random(base + k))
            lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 24, 0
            lim.mbox_regs, lim.mbox_msgs = 4, 6
            _compact(lim)
            try:
                parity.gpu_compare(hip, w, k * 5, 96, cfg, lim, "compact:random_workload", TALLY,
                                   (f"random_workload(Random({base + k})) [MADSIM_FUZZ_SEED={FUZZ_SEED}]", desc))
            except RuntimeError:
                continue                                 # not a compact candidate (tasks, horizon, buggify)
            ran += 1
    assert ran >= 30, ran


# ---- MADSIM_STATE_DEDUP_TIMERS: re-registered Sleep timers as counts (k_timer.h dedup_note) ---------------------------------

def test_dedup_timers_election_loop_gpu(hip):
    """The election loop at batch size with the repeats of a pending Sleep's timer kept as counts: the same 48 bytes per seed
    as without the switch (131 072 seeds compared with each other), sampled and contiguous seeds against the oracle."""
    from tests import lifecycle_workloads as LW
    w, lim = W.raft_election(), W.raft_election_limits()
    lim.state_mem = A.STATE_GLOBAL
    g0, g1 = hip.geometry(w, lim), hip.geometry(w, LW.dedup_limits(lim))
    assert g0.variant == g1.variant and g1.global_bytes_per_seed == g0.global_bytes_per_seed + 64 * 16
    plain, _ = hip.run_batch(w, 0, 131072, None, lim)
    got, _ = hip.run_batch(w, 0, 131072, None, LW.dedup_limits(lim))
    assert (got == plain).all()
    want, _ = oracle.run_batch(w, 40000, 1024, None, lim)
    assert (got[40000:41024] == want).all()
    for s in [(k * 4093) % 131072 for k in range(128)]:
        want, _ = oracle.run_batch(w, s, 1, None, lim)
        assert got[s] == want[0], f"seed {s}"


def test_dedup_timers_ties_and_lane_reuse_gpu(hip):
    """Ties between different events (a seventh of these seeds): the kernel restarts such a seed with the literal heap — also
    when the lane has more seeds to run afterwards (the per-launch work queue and static striding, more seeds than lanes)."""
    from tests import lifecycle_workloads as LW
    w = LW.timeout_repeats_and_ties()
    lim = A.Limits(); lim.heap_lds_slots, lim.heap_spill_slots = 4, 60; lim.mbox_regs, lim.mbox_msgs = 8, 8
    want, _ = oracle.run_batch(w, 0, 8192, None, lim)
    got, _ = hip.run_batch(w, 0, 8192, None, LW.dedup_limits(lim))
    assert (got == want).all() and (got["verdict"] == A.PASS).all()
    n = 3 * hip.geometry(w, LW.dedup_limits(lim)).grid_blocks * 256 // 2 + 777       # one and a half seeds per resident lane ... and a few
    n = min(n, 400000)
    plain = copy_limits(lim); plain.lanes_per_wave, plain.state_mem = 0, A.STATE_GLOBAL
    ref, _ = hip.run_batch(w, 1 << 20, n, None, plain)
    for sched in (A.SCHED_STATIC, A.SCHED_QUEUE):
        l2 = LW.dedup_limits(lim); l2.sched = sched
        got, _ = hip.run_batch(w, 1 << 20, n, None, l2)
        assert (got == ref).all(), sched
    want, _ = oracle.run_batch(w, (1 << 20) + n - 2048, 2048, None, lim)
    assert (ref[n - 2048:] == want).all()


# ---- MADSIM_STATE_NARROW_HEAP: 8-byte timer-heap entries + delivery record pool (k_timer.h nh_*, round 6) -------------------------

def _narrow(lim, heap_lds=None):
    l2 = copy_limits(lim) if lim is not None else A.Limits()
    l2.state_mem = (l2.state_mem if (l2.state_mem & 0xff) else A.STATE_GLOBAL) | A.STATE_NARROW_HEAP
    if l2.lanes_per_wave != 32:
        l2.lanes_per_wave = 0
    if heap_lds is not None:
        l2.heap_spill_slots, l2.heap_lds_slots = l2.heap_spill_slots + max(0, l2.heap_lds_slots - heap_lds), heap_lds
    return l2


@pytest.mark.parametrize("name,quotas", [("topo", (15, 31)), ("raft", (8, 20))])
def test_narrow_heap_bench_workloads_gpu(hip, name, quotas):
    """configs[2] / configs[4] shapes on 8-byte heap entries (what bench.py runs since round 6): the same 48 bytes per seed as on the 16-byte
    entries (65 536 seeds compared with each other at two LDS quotas), contiguous and sampled seeds against the oracle, packet loss."""
    w, lim, _ = W.bench_case(name)
    assert (hip.geometry(w, lim).variant >> 8) & 0x80, "the bench case runs the narrow-heap build"
    wlim = copy_limits(lim); wlim.state_mem &= ~A.STATE_NARROW_HEAP
    assert not (hip.geometry(w, wlim).variant >> 8) & 0x80
    wide, n_ovf = parity.run_resolved(hip, w, 0, 65536, None, wlim)
    for q in quotas:
        l2 = _narrow(lim, q)
        assert (hip.geometry(w, l2).variant >> 8) & 0x80, "the narrow-heap build was not selected"
        got, _ = parity.run_resolved(hip, w, 0, 65536, None, l2)
        assert (got == wide).all(), q
    want, _ = oracle.run_batch(w, 50000, 768, None, lim)
    assert (wide[50000:50768] == want).all()
    _cmp(hip, w, 8_100_000, 1024, A.Config.default(packet_loss_rate=0.04), _narrow(hip.grow_limits(lim), quotas[0]))


def test_narrow_heap_reruns_leave_the_layout_gpu(hip):
    """The two run-time conditions the 8-byte entries cannot hold — a channel back-off past the 2^31 ns horizon, more datagrams in flight than the
    record pool has records — are capacity verdicts of the plain call, and `madsim_hip_run_batch_auto` answers them on the 16-byte entries with
    the oracle's bytes (madsim_hip.cpp grow() clears MADSIM_STATE_NARROW_HEAP)."""
    from tests import lifecycle_workloads as LW
    w, lim = LW.narrow_heap_backoff_past_the_horizon()
    l2 = _narrow(lim)
    assert (hip.geometry(w, l2).variant >> 8) & 0x80
    first, _ = hip.run_batch(w, 0, 512, None, l2)
    assert (first["verdict"] == A.OVERFLOW).all()
    parity.gpu_compare(hip, w, 0, 512, None, l2, "narrow horizon", TALLY, "narrow horizon")
    w, cfg, lim = LW.narrow_heap_forty_datagrams_in_flight()
    l2 = _narrow(lim)
    first, _ = hip.run_batch(w, 0, 512, cfg, l2)
    assert (first["verdict"] == A.OVERFLOW).any()
    parity.gpu_compare(hip, w, 0, 512, cfg, l2, "narrow pool", TALLY, "narrow pool")


def test_narrow_heap_fuzz_gpu(hip):
    """Random timeout-only / latency / mixed / RPC programs on 8-byte heap entries with small LDS quotas (pushes and pops walk the spilled
    levels; deliveries ride the record pool; a deadline beyond the horizon or a full pool is a capacity verdict, re-run wide)."""
    from tests import fuzz

    def limits_of(base, quota):
        def f():
            f.k += 1
            lim = _narrow(base(), quota[f.k % len(quota)])
            if base is fuzz.mailbox_limits and f.k % 3 == 0:
                lim.state_mem |= A.STATE_DEDUP_TIMERS
            return lim
        f.k = 0
        return f
    _fuzz_two_blocks(hip, fuzz.random_timeout_workload, 66100, 120, 60, 21, count=64, seed_mul=13, limits=limits_of(fuzz.mailbox_limits, (1, 2, 3, 5, 8)))
    _fuzz_two_blocks(hip, fuzz.random_latency_workload, 66300, 100, 50, 22, count=64, seed_mul=5, limits=limits_of(fuzz.mailbox_limits, (1, 3, 8)))
    _fuzz_two_blocks(hip, fuzz.random_mixed_workload, 66500, 100, 50, 23, count=64, seed_mul=3, limits=limits_of(fuzz.mixed_limits, (2, 4, 8)))
    _fuzz_two_blocks(hip, fuzz.random_rpc_workload, 66700, 80, 40, 24, count=64, seed_mul=7, limits=limits_of(fuzz.mailbox_limits, (1, 2, 6)))


def copy_limits(lim):
    import copy
    return copy.copy(lim)


def test_fuzz_timeout_workloads_gpu(hip):
    """Random timeout-only programs (repeats of pending Sleeps, nanosecond ties, sleep_until / advance, partitions) on the
    de-duplicating global-state build — and, every other program, on the builds the switch does not touch."""
    from tests import fuzz, lifecycle_workloads as LW

    def limits():
        limits.k += 1
        lim = fuzz.mailbox_limits()
        return LW.dedup_limits(lim) if limits.k % 2 else lim
    limits.k = 0
    _fuzz_two_blocks(hip, fuzz.random_timeout_workload, 12100, 240, 120, 11, count=64, seed_mul=13, limits=limits)


def test_fuzz_reply_without_receive_gpu(hip):
    """`reply` with an unset or stale `from` (tests/fuzz.py random_reply_without_receive_workload): entry 0 on both sides."""
    from tests import fuzz
    _fuzz_two_blocks(hip, fuzz.random_reply_without_receive_workload, 13300, 120, 60, 12, count=32, seed_mul=3, limits=fuzz.mailbox_limits, alt_global=True)


def test_fuzz_unstructured_workloads_gpu(hip):
    """Op soup (tests/fuzz.py random_unstructured_workload) on the GPU: the oracle's answer or a refusal, in both state layouts."""
    import random
    from tests import fuzz
    from madsim_amd import runtime
    for base, n in ((14400, 200), ((FUZZ_SEED * 1_000_003 + 13 * 7919) & 0x7fffffffffff, 100)):
        for k in range(n):
            w, cfg, desc = fuzz.random_unstructured_workload(random.Random(base + k))
            lim = fuzz.generous_limits(); lim.max_tasks = 16
            if k % 2:
                lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS
            try:
                parity.gpu_compare(hip, w, k * 3, 24, cfg, lim, "random_unstructured_workload", TALLY,
                                   (f"random_unstructured_workload(Random({base + k})) [MADSIM_FUZZ_SEED={FUZZ_SEED}]", desc))
            except runtime.MadsimHipError:                  # refused by validate()
                continue


def test_fuzz_unstructured_wide_gpu(hip):
    """The op soup over the whole table format (tests/fuzz.py random_unstructured_wide_workload: address kinds, port-0 entries,
    IP-less nodes, typed RPC, hooks, IPVS calls, panics on restarting nodes) on the GPU, both state layouts, a fixed block and a
    fresh one: the oracle's 48 bytes, a capacity verdict or a refusal; MADSIM_UNSUPPORTED exactly where the oracle says it; never
    MADSIM_INTERNAL."""
    import random
    from tests import fuzz
    from madsim_amd import runtime
    n_unsup = 0
    for base, n in ((1_500_000, 300), ((FUZZ_SEED * 1_000_003 + 17 * 7919) & 0x7fffffffffff, 150)):
        for k in range(n):
            w, cfg, desc = fuzz.random_unstructured_wide_workload(random.Random(base + k))
            lim = fuzz.wide_limits(k % 2)
            try:
                got, _ = parity.gpu_compare(hip, w, k * 3, 24, cfg, lim, "random_unstructured_wide_workload", TALLY,
                                            (f"random_unstructured_wide_workload(Random({base + k})) [MADSIM_FUZZ_SEED={FUZZ_SEED}]", desc))
            except runtime.MadsimHipError:                  # refused by validate()
                continue
            assert not (got["verdict"] == A.INTERNAL).any()
            n_unsup += int((got["verdict"] == A.UNSUPPORTED).sum())
    assert n_unsup > 0


def test_restarted_nodes_leftover_socket_same_bytes_with_and_without_connection_ops_gpu(hip):
    """kernel builds with and without connection ops disagreed on a datagram for the socket a restarted node's
    dead task left in the table (net/mod.rs:483-493, task/mod.rs:374-401).  Same program, once with a stray `drop((tx, rx))` that
    selects the connection build: identical bytes from both builds, both layouts, and the oracle's."""
    from tests import fuzz
    from madsim_amd import workload as W
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
            want, _ = oracle.run_batch(w, 0, 256, None, lim)
            got, _ = hip.run_batch(w, 0, 256, None, lim)
            assert (got == want).all(), (with_conn, glob)
            outs.append(got)
    for f in ("verdict", "steps", "clock_ns", "msg_count", "rng_calls", "trace_hash", "obs_hash"):
        assert (outs[0][f] == outs[2][f]).all() and (outs[1][f] == outs[3][f]).all(), f


def test_port0_rebind_is_unsupported_and_never_rerun_gpu(hip):
    """`bind; bind` on a port-0 entry: MADSIM_UNSUPPORTED from kernel and oracle, every other field 0; run_batch_auto does not
    re-run it (larger limits change nothing) and counts it as a runner verdict, not a genuine failure."""
    from tests import fuzz
    from madsim_amd import workload as W
    wl = W.WorkloadBuilder(); n = wl.create_node()
    a = wl.addr(n, 0, ip="unspecified")
    t = wl.task(n); t.bind(a); t.bind(a); t.done()
    m = wl.main(); m.spawn(t); m.join(t); m.done()
    w = wl.build()
    for glob in (0, 1):
        lim = fuzz.wide_limits(glob)
        want, _ = oracle.run_batch(w, 0, 64, None, lim)
        got, _ = hip.run_batch(w, 0, 64, None, lim)
        assert (got == want).all() and (got["verdict"] == A.UNSUPPORTED).all() and not got["steps"].any()
        got2, _ = hip.run_batch_auto(w, 0, 64, None, lim)
        assert (got2 == want).all()


def test_run_batch_of_262144_seeds_is_pipelined_and_bit_exact(hip):
    """VERDICT r3 weak #5 / Builder::run (runtime/builder.rs:121-162: one call runs all seeds): a plain madsim_hip_run_batch with
    more than one batch of seeds is cut into sub-launches kept in flight inside the library and still fills the caller's
    per-seed array — identical to four separate one-batch calls, to the oracle on sampled seeds (all 48 bytes) and in its
    summary; a ragged tail, loss-induced failures (first failing seed), and the compact / plain / global layouts."""
    from madsim_amd import workload
    w, lim, _ = workload.bench_case("pingpong")
    for cfg, count in ((None, 262144), (A.Config.default(packet_loss_rate=2e-6), 200_001)):
        got, summ = hip.run_batch(w, 77_000_000, count, cfg, lim)
        parts, nf, steps, first = [], 0, 0, (1 << 64) - 1
        for lo in range(0, count, 65536):
            g1, s1 = hip.run_batch(w, 77_000_000 + lo, min(65536, count - lo), cfg, lim)
            parts.append(g1); nf += s1.n_failed; steps += s1.total_steps; first = min(first, s1.first_failing_seed)
        assert (got == np.concatenate(parts)).all()
        assert (summ.n_failed, summ.total_steps, summ.first_failing_seed) == (nf, steps, first)
        idx = (np.arange(768) * 341) % count
        want = np.concatenate([oracle.run_batch(w, 77_000_000 + int(i), 1, cfg, lim)[0] for i in idx])
        assert (got[idx] == want).all()
        if cfg is not None:
            assert summ.n_failed > 0 and got[summ.first_failing_seed - 77_000_000]["verdict"] == A.DEADLOCK
    # an extended-op workload in the global layout with a heap-spill region (scratch per stream), 3 sub-batches + a tail
    xw, xlim, _ = workload.bench_case("kv")
    got, summ = hip.run_batch(xw, 5_000_000, 150_000, None, xlim)
    one, _ = hip.run_batch(xw, 5_000_000 + 131072, 150_000 - 131072, None, xlim)
    assert (got[131072:] == one).all()
    res, n_ovf = parity.run_resolved(hip, xw, 5_000_000, 150_000, None, xlim)       # (the pipelined first pass again, then the re-run of its capacity verdicts)
    assert ((res == got) | (got["verdict"] == A.OVERFLOW)).all() and n_ovf < 0.005 * 150_000
    idx = (np.arange(256) * 577) % 150_000
    want = np.concatenate([oracle.run_batch(xw, 5_000_000 + int(i), 1, None, xlim)[0] for i in idx])
    assert (res[idx] == want).all()


def test_run_batch_multi_over_k_contexts_equals_k_independent_calls(hip):
    """madsim_hip_run_batch_multi over k contexts == k independent run_batch calls on the blocks of madsim_amd/dist.py
    shard_range, for counts k does not divide and counts large enough that every context pipelines several sub-batches
    (runtime/builder.rs:129-150: the seeds are seed0 .. seed0 + count whatever runs them)."""
    from madsim_amd import dist as mdist
    w = W.pingpong(4, 8)
    cfg = A.Config.default(packet_loss_rate=0.002)
    for k, count in ((3, 10_007), (5, 65_537), (2, 300_001), (4, 3)):
        ctxs = [hip.Context(0) for _ in range(k)]
        try:
            got, summ = hip.run_batch_multi(ctxs, w, 900, count, cfg)
            parts, nf, first = [], 0, (1 << 64) - 1
            for g in range(k):
                lo, n = mdist.shard_range(900, count, g, k)
                g1, s1 = ctxs[g].run_batch(w, lo, n, cfg)
                parts.append(g1); nf += s1.n_failed
                if s1.n_failed:
                    first = min(first, s1.first_failing_seed)
            assert (got == np.concatenate(parts)).all(), (k, count)
            assert (summ.n_failed, summ.first_failing_seed) == (nf, first)
        finally:
            for c in ctxs:
                c.close()
    want, _ = oracle.run_batch(w, 900, 3000, cfg)
    ctxs = [hip.Context(0) for _ in range(3)]
    try:
        got, _ = hip.run_batch_multi(ctxs, w, 900, 3000, cfg)
    finally:
        for c in ctxs:
            c.close()
    assert (got == want).all()


# ---- round 4: 32 seed lanes per wave on the timeout-only global-state build (the election loop's bench case) -----------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["receiver_drop", "request_timeout_with_stale_timers", "dead_registrations_swept_by_delivery",
                                  "many_endpoints_dropped_in_table_order"])
def test_global_state_with_32_seed_lanes_per_wave_gpu(hip, name):
    lim = LW.limits(name) or A.Limits()
    lim.lanes_per_wave, lim.state_mem = 32, A.STATE_GLOBAL
    g = hip.geometry(LW.ALL[name](), lim)
    assert g.lanes_per_wave == 32 and g.global_bytes_per_seed > 0 and (g.variant >> 16) & 0xf == 5
    _cmp(hip, LW.ALL[name](), 0, 1024, LW.config(name), lim)
    lim.state_mem |= A.STATE_DEDUP_TIMERS
    _cmp(hip, LW.ALL[name](), 5000, 1024, A.Config.default(packet_loss_rate=0.05), lim)


@pytest.mark.gpu
def test_election_loop_32_and_64_seed_lanes_give_the_same_bytes_gpu(hip):
    """Round 4's layout of the election loop (32 seed lanes per wave, 22 wide heap entries in LDS) against the same workload on full waves (the
    bench case since round 6, on 8-byte heap entries): 65 536 seeds compared with each other, 1 024 contiguous and 128 scattered ones with
    the oracle."""
    w, full = W.raft_election(), W.raft_election_limits()
    g = hip.geometry(w, full)
    assert full.lanes_per_wave == 0 and g.lanes_per_wave == 64 and (g.variant >> 8) & 0x80
    lim = W.raft_election_limits(); lim.lanes_per_wave = 32; lim.state_mem = A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS
    g = hip.geometry(w, lim)
    assert g.lanes_per_wave == 32 and (g.variant >> 16) & 0xf == 5 and not (g.variant >> 8) & 0x80
    a, na = parity.run_resolved(hip, w, 0, 65536, None, lim)
    b, nb = parity.run_resolved(hip, w, 0, 65536, None, full)
    assert (a == b).all() and na + nb < 0.01 * 65536, (na, nb)
    want, _ = oracle.run_batch(w, 20000, 1024, None, lim)
    assert (a[20000:21024] == want).all()
    for s in [(k * 4093) % 65536 for k in range(128)]:
        want, _ = oracle.run_batch(w, s, 1, None, lim)
        assert a[s] == want[0], f"seed {s}"


def test_summary_only_call_with_null_result_array(hip):
    """`out` may be NULL when only the summary is wanted (include/madsim_hip.h, madsim_hip_run_batch) — for one launch and for a
    count the library cuts into sub-launches (the pipelined branch refused it in round 4: ADVICE r4)."""
    import ctypes as C
    L = hip.lib()
    w = W.pingpong(4, 8)
    cfg, lim = A.Config.default(packet_loss_rate=0.002), A.Limits()
    for count in (1, 4096, 262144 + 77):
        got, want = hip.run_batch(w, 9_000_000, count, cfg, lim)
        s = A.Summary()
        rc = L.madsim_hip_run_batch(w.ref(), C.byref(cfg), 9_000_000, count, C.byref(lim), None, C.byref(s))
        assert rc == 0, L.madsim_hip_last_error()
        assert (s.n_failed, s.first_failing_seed, s.total_steps, s.total_clock_ns) == \
               (want.n_failed, want.first_failing_seed, want.total_steps, want.total_clock_ns)
        assert s.n_failed == int((got["verdict"] != A.PASS).sum())


def test_campaign_over_several_contexts_reports_like_one_and_stops_everywhere(hip):
    """madsim_hip_run_campaign_multi (runtime/builder.rs:129-160 across devices): batch k on context k % n, reports read in batch
    order.  k contexts (all on GPU 0 here: the code path of one per GPU) give the report of ONE context for the same range — no stop:
    every count and the oracle's first failing seed; STOP_AT_FAILURE: the same first failing seed after reading exactly the batches up
    to it, at most one round of n x in_flight batches launched beyond; ragged ranges, fewer batches than contexts, one batch."""
    w = W.pingpong(4, 16)
    FIELDS = ("seeds_run", "batches_run", "first_failing_seed", "n_failed", "n_runner", "total_steps", "total_clock_ns")
    with hip.Context(0) as c0, hip.Context(0) as c1, hip.Context(0) as c2:
        for ctxs in ([c0, c1], [c0, c1, c2], [c2]):
            cfg = A.Config.default(packet_loss_rate=0.002)
            for total, batch in ((40_000, 4096), (4096 * 7 + 5, 4096), (1000, 4096), (4096 * 2, 4096)):
                one = hip.run_campaign(w, 5_000_000, total, batch, 3, False, cfg)
                many = hip.run_campaign_multi(ctxs, w, 5_000_000, total, batch, 2, False, cfg)
                assert [getattr(many, f) for f in FIELDS] == [getattr(one, f) for f in FIELDS], (len(ctxs), total)
                assert many.batches_launched == many.batches_run == (total + batch - 1) // batch
            _, osum = oracle.run_batch(w, 5_000_000, 40_000, cfg)
            assert (many.n_failed, many.first_failing_seed) != (0, (1 << 64) - 1)
            many = hip.run_campaign_multi(ctxs, w, 5_000_000, 40_000, 4096, 2, False, cfg)
            assert (many.n_failed, many.first_failing_seed, many.total_steps) == (osum.n_failed, osum.first_failing_seed, osum.total_steps)
            # rare failures: every device stops within one round of the batch that holds the first one
            cfg = A.Config.default(packet_loss_rate=0.000002)
            batch = 4096
            one = hip.run_campaign(w, 9_000_000, 64 * batch, batch, 3, True, cfg)
            many = hip.run_campaign_multi(ctxs, w, 9_000_000, 64 * batch, batch, 2, True, cfg)
            assert many.first_failing_seed == one.first_failing_seed != (1 << 64) - 1
            j = (many.first_failing_seed - 9_000_000) // batch
            assert [getattr(many, f) for f in FIELDS] == [getattr(one, f) for f in FIELDS]
            assert many.batches_run == j + 1 and j + 1 <= many.batches_launched <= j + 1 + 2 * len(ctxs)
        # the same context twice / none: refused, nothing launched
        with pytest.raises(hip.MadsimHipError):
            hip.run_campaign_multi([c0, c0], w, 0, 100)
        with pytest.raises(hip.MadsimHipError):
            hip.run_campaign_multi([], w, 0, 100)


def test_campaign_over_ranks_on_one_gpu_equals_the_library_campaign(hip):
    """madsim_amd/runtime.py run_campaign_over_ranks without a process group (world 1; the multi-rank fold is tested over gloo on the CPU,
    tests/test_dist_gloo.py): the same report as madsim_hip_run_campaign for the same prefix, with and without the early stop — one batch
    per round (round 5's form) and pipelined rounds (round 6: each round one campaign call that keeps its batches in flight)."""
    w = W.pingpong(4, 16)
    for loss, total, stop in ((0.002, 40_000, False), (0.000002, 64 * 4096, True)):
        cfg = A.Config.default(packet_loss_rate=loss)
        one = hip.run_campaign(w, 9_000_000, total, 4096, 3, stop, cfg)
        for rb in (1, 0, 7):
            got = hip.run_campaign_over_ranks(w, 9_000_000, total, 4096, stop, cfg, round_batches=rb)
            assert (got["first_failing_seed"], got["n_failed"], got["n_runner"], got["total_steps"], got["seeds_run"], got["batches_run"]) == \
                   (one.first_failing_seed, one.n_failed, one.n_runner, one.total_steps, one.seeds_run, one.batches_run), rb


def test_campaign_over_ranks_runs_at_the_pipelined_rate(hip):
    """the multi-process seed search must not run one launch at a time.  Headline workload, 65 536-seed batches: the rate of
    run_campaign_over_ranks (default rounds: four times the batches in flight per call) against ONE madsim_hip_run_campaign call over the
    same seeds — within a few percent (a round's ramp and drain, one 56-byte exchange); round 5's one-batch rounds for comparison."""
    import time
    w, lim, _ = W.bench_case("pingpong")
    n = 120 * 65536
    hip.run_campaign(w, 1 << 30, 10 * 65536, 65536, 0, False, None, lim)            # warm
    best = {}
    for _ in range(3):
        t = time.perf_counter(); one = hip.run_campaign(w, 1 << 31, n, 65536, 0, False, None, lim); d1 = time.perf_counter() - t
        t = time.perf_counter(); got = hip.run_campaign_over_ranks(w, 1 << 31, n, 65536, False, None, lim); d2 = time.perf_counter() - t
        t = time.perf_counter(); hip.run_campaign_over_ranks(w, 1 << 31, n // 4, 65536, False, None, lim, round_batches=1); d3 = (time.perf_counter() - t) * 4
        for k, d in (("one", d1), ("ranks", d2), ("ranks_1", d3)):
            best[k] = min(best.get(k, d), d)
        assert (got["n_failed"], got["total_steps"], got["seeds_run"]) == (one.n_failed, one.total_steps, one.seeds_run)
    print(f"campaign {n / best['one'] / 1e6:.1f} M seeds/s, over ranks (pipelined rounds) {n / best['ranks'] / 1e6:.1f}, one batch per round {n / best['ranks_1'] / 1e6:.1f}")
    assert best["ranks"] <= 1.07 * best["one"], best


