"""The C-ABI boundary: the library loads, exports every symbol include/madsim_hip.h declares, its structs
have the documented layout, and it fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

from madsim_amd import _abi as A
from madsim_amd import runtime, workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "madsim_hip.h")).read()


def test_library_exports_every_declared_symbol():
    L = runtime.lib()
    names = set(re.findall(r"\b(madsim_(?:hip|workload)_[a-z_]+)\s*\(", HEADER))
    assert {"madsim_hip_run_batch", "madsim_hip_run_batch_device", "madsim_hip_trace_seed", "madsim_hip_init",
            "madsim_hip_shutdown", "madsim_hip_version", "madsim_hip_geometry", "madsim_workload_pingpong", "madsim_hip_run_batch_auto",
            "madsim_hip_strerror", "madsim_hip_last_error"} <= names
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/madsim_hip.h but not exported"
    assert L.madsim_hip_version() == A.ABI_VERSION == int(re.search(r"MADSIM_HIP_ABI_VERSION (\d+)u", HEADER).group(1))


def test_struct_layouts_match_header():
    """The ctypes mirror against the header itself: every struct's field names, order, offsets and size (parsed from
    include/madsim_hip.h by tests/cheader.py), every op code."""
    from tests import cheader as H
    S = H.structs()
    mirror = {"madsim_insn_t": A.Insn, "madsim_prog_t": A.Prog, "madsim_sock_t": A.Sock, "madsim_node_t": A.Node,
              "madsim_workload_t": A.Workload, "madsim_config_t": A.Config, "madsim_limits_t": A.Limits,
              "madsim_result_t": A.Result, "madsim_summary_t": A.Summary, "madsim_geometry_t": A.Geometry}
    for name in S:
        if name in mirror:
            continue
        assert hasattr(A, "HEADER_STRUCTS") and name in A.HEADER_STRUCTS, f"{name} has no ctypes mirror in madsim_amd/_abi.py"
    mirror.update(getattr(A, "HEADER_STRUCTS", {}))
    for name, cls in mirror.items():
        offs, size = H.layout(S[name])
        assert [f[0] for f in cls._fields_] == [f[0] for f in S[name]], name
        assert C.sizeof(cls) == size, (name, C.sizeof(cls), size)
        for fname, _, _, _ in S[name]:
            assert getattr(cls, fname).offset == offs[fname], (name, fname)
    assert C.sizeof(A.Result) == 48 and A.Result.trace_hash.offset == 32 and A.Result.obs_hash.offset == 40
    for name, val in A.OP.items():
        m = re.search(r"MS_OP_%s = (\d+)," % name, HEADER)
        assert m and int(m.group(1)) == val, name
    assert len(A.OP) == len(re.findall(r"MS_OP_\w+ = \d+,", HEADER))


def test_c_twin_of_pingpong_matches_python_assembler():
    L = runtime.lib()
    for n, r in ((2, 64), (4, 64), (16, 3)):
        nodes, progs, socks = (A.Node * (n + 1))(), (A.Prog * (n + 1))(), (A.Sock * n)()
        insns = (A.Insn * 512)()
        w = A.Workload()
        cnt = L.madsim_workload_pingpong(n, r, nodes, progs, socks, insns, 512, C.byref(w))
        py = workload.pingpong(n, r)
        assert cnt == py.struct.n_insns == w.n_insns
        assert bytes(insns)[:cnt * 8] == bytes(py.insns)
        assert bytes(progs) == bytes(py.progs) and bytes(socks) == bytes(py.socks)
    assert L.madsim_workload_pingpong(3, 1, nodes, progs, socks, insns, 512, C.byref(w)) == -1   # odd node count


def test_geometry_and_validation_need_no_gpu():
    g = runtime.geometry(workload.pingpong(4, 64))
    assert g.block_threads in (64, 128, 256) and g.lanes_per_wave == 64 and g.lds_bytes_per_seed > 0
    lim = A.Limits(); lim.lanes_per_wave = 7
    with pytest.raises(runtime.MadsimHipError, match="lanes_per_wave"):
        runtime.geometry(workload.pingpong(4, 64), lim)
    bad = workload.WorkloadBuilder(); bad.main().jmp(500)
    with pytest.raises(runtime.MadsimHipError, match="jump target"):
        runtime.geometry(bad.build())


def test_program_table_validation():
    """What the kernel relies on instead of checking at run time: the instruction table ends in a terminator (no body can run off
    its end), every jump target is inside it; and the rules of MADSIM_PROG_DROP_SPAWN (the guard spawns the NEXT program, on the
    same node, never together with pause) — refused by the library and by the oracle alike."""
    import oracle
    def raw(insns, progs, nodes=1):
        w = workload.WorkloadBuilder()
        for _ in range(nodes):
            w.create_node()
        return workload.BuiltWorkload(w.nodes, progs, [], insns)
    off_end = raw([A.Insn(A.OP["SLEEP"], 0, 0, 5)], [A.Prog(0, 0, 0)])                      # ... and then what?
    with pytest.raises(runtime.MadsimHipError, match="must end in"):
        runtime.geometry(off_end)
    with pytest.raises(RuntimeError):
        oracle.run_batch(off_end, 0, 1)
    bad_jeq = raw([A.Insn(A.OP["JEQ"], 0, 9, 0), A.Insn(A.OP["DONE"], 0, 0, 0)], [A.Prog(0, 0, 0)])
    with pytest.raises(runtime.MadsimHipError, match="jump target"):
        runtime.geometry(bad_jeq)
    with pytest.raises(RuntimeError):
        oracle.run_batch(bad_jeq, 0, 1)
    # a guard must have a next program, on the guard's node ...
    last = workload.WorkloadBuilder(); n = last.create_node(); last.task(n, spawn_on_drop=True)
    with pytest.raises(runtime.MadsimHipError, match="DROP_SPAWN"):
        runtime.geometry(last.build())
    with pytest.raises(RuntimeError):
        oracle.run_batch(last.build(), 0, 1)
    other = workload.WorkloadBuilder(); n1, n2 = other.create_node(), other.create_node()
    other.task(n1, spawn_on_drop=True); other.task(n2)
    with pytest.raises(runtime.MadsimHipError, match="same node"):
        runtime.geometry(other.build())
    # ... and no pause anywhere in the workload (a parked Runnable is dropped in the KILLER's context)
    paused = workload.WorkloadBuilder(); n = paused.create_node()
    paused.task(n, spawn_on_drop=True); paused.task(n); paused.main().pause(n)
    with pytest.raises(runtime.MadsimHipError, match="MS_OP_PAUSE"):
        runtime.geometry(paused.build())
    with pytest.raises(RuntimeError):
        oracle.run_batch(paused.build(), 0, 1)
    ok = workload.WorkloadBuilder(); n = ok.create_node(); ok.task(n, spawn_on_drop=True); ok.task(n)
    g = runtime.geometry(ok.build())
    assert (g.variant >> 8) & 0x1f == 31 or "ALL" in runtime.variant_name(g)           # guards live in the full builds


def test_ephemeral_endpoints_validation():
    """port 0 = an ephemeral Endpoint (network.rs:224-236): accepted, picks the full-address build, counts its candidate
    ports against the 63-entry table, and cannot be named as a destination — by the library and by the oracle alike."""
    import oracle
    wl = workload.WorkloadBuilder()
    n = wl.create_node(); eph, named = wl.addr(n, 0, ip="unspecified"), wl.addr(n, 7)
    t = wl.task(n); t.bind(eph, port_to_val=True); t.assert_val(1); t.send_to(eph, named, 1, 5); t.done()
    m = wl.main(); m.spawn(t); m.join(t)
    g = runtime.geometry(wl.build())
    assert "ALL" in runtime.variant_name(g) or (g.variant >> 8) & 0x1f == 31
    bad = workload.WorkloadBuilder()
    n = bad.create_node(); eph, named = bad.addr(n, 0), bad.addr(n, 7)
    t = bad.task(n); t.bind(named); t.send_to(named, eph, 1, 5); t.done()
    bad.main().spawn(t)
    with pytest.raises(runtime.MadsimHipError, match="ephemeral"):
        runtime.geometry(bad.build())
    with pytest.raises(RuntimeError):
        oracle.run_batch(bad.build(), 0, 1)
    many = workload.WorkloadBuilder()
    n = many.create_node()
    ephs = [many.addr(n, 0, ip="unspecified") for _ in range(8)]      # 8 handles + 8 candidate ports each group = 16: fine
    t = many.task(n)
    for e in ephs:
        t.bind(e)
    t.done(); many.main().spawn(t)
    runtime.geometry(many.build())
    for _ in range(24):                                                 # 32 handles + 32 candidates > 63
        many.addr(n, 0, ip="unspecified")
    with pytest.raises(runtime.MadsimHipError, match="63 socket addresses"):
        runtime.geometry(many.build())


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(runtime.MadsimHipError):
        runtime.init(0)
    L = runtime.lib()
    cfg, lim, summ = A.Config.default(), A.Limits(), A.Summary()
    w = workload.pingpong(2, 1)
    rc = L.madsim_hip_run_batch(w.ref(), C.byref(cfg), 0, 1, C.byref(lim), None, C.byref(summ))
    assert rc == -3 and b"not initialised" in L.madsim_hip_strerror(rc)        # MADSIM_E_NOINIT


def test_builder_from_env_mirrors_reference(tmp_path):
    """runtime/builder.rs:64-118"""
    b = runtime.Builder.from_env({"MADSIM_TEST_SEED": "42", "MADSIM_TEST_NUM": "7", "MADSIM_TEST_JOBS": "3",
                                  "MADSIM_TEST_TIME_LIMIT": "1.5"})
    assert (b.seed, b.count, b.jobs, b.time_limit, b.check) == (42, 7, 3, 1.5, False)
    assert b.limits().time_limit_ns == 1_500_000_000
    b = runtime.Builder.from_env({"MADSIM_TEST_CHECK_DETERMINISM": "1"})
    assert b.check and b.count == 2 and b.jobs == 1 and 0 < b.seed < 2**64
    with pytest.raises(ValueError, match="MADSIM_TEST_SEED should be an integer"):
        runtime.Builder.from_env({"MADSIM_TEST_SEED": "x"})
    cfgfile = tmp_path / "c.toml"
    cfgfile.write_text('[net]\npacket_loss_rate = 0.25\nsend_latency = { start = { secs = 0, nanos = 2000000 }, '
                       'end = { secs = 0, nanos = 3000000 } }\n')
    b = runtime.Builder.from_env({"MADSIM_TEST_CONFIG": str(cfgfile)})
    assert (b.config.packet_loss_rate, b.config.lat_lo_ns, b.config.lat_hi_ns) == (0.25, 2_000_000, 3_000_000)


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu():
    """include/madsim_hip.hpp + examples/pingpong_test.cpp: the C++ Builder::from_env().run(workload)."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "examples", "pingpong_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "examples", "pingpong_test.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "madsim_amd"), "-lmadsim_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "madsim_amd")])
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the run itself is covered by the gpu test")
    p = subprocess.run([exe], env=dict(os.environ, MADSIM_TEST_SEED="1", MADSIM_TEST_NUM="4"), capture_output=True, text=True)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr


def test_oracle_exports_the_cpu_twin_with_the_product_signature():
    """madsim_cpu_run_batch(...) with the identical signature of madsim_hip_run_batch."""
    import numpy as np
    import oracle
    L = oracle.lib()
    L.madsim_cpu_run_batch.argtypes = runtime.lib().madsim_hip_run_batch.argtypes
    w = workload.pingpong(2, 4)
    cfg, lim, summ = A.Config.default(), A.Limits(), A.Summary()
    out = np.zeros(16, dtype=A.RESULT_DTYPE)
    assert L.madsim_cpu_run_batch(w.ref(), C.byref(cfg), 3, 16, C.byref(lim), out.ctypes.data_as(C.c_void_p), C.byref(summ)) == 0
    want, osm = oracle.run_batch(w, 3, 16)
    assert (out == want).all() and summ.total_steps == osm.total_steps


def test_builder_rejects_what_rust_types_reject():
    """Builder fields are u64 / u64 / u16 (builder.rs:7-22); seed + i must not wrap (builder.rs:129)."""
    with pytest.raises(ValueError):
        runtime.Builder(seed=-1)
    with pytest.raises(ValueError):
        runtime.Builder(seed=2**64)
    with pytest.raises(ValueError):
        runtime.Builder(seed=2**64 - 2, count=3)
    runtime.Builder(seed=2**64 - 2, count=2)
    with pytest.raises(ValueError):
        runtime.Builder(jobs=70000)
    with pytest.raises(ValueError, match="MADSIM_TEST_SEED"):
        runtime.Builder.from_env({"MADSIM_TEST_SEED": "-5"})
    with pytest.raises(ValueError, match="MADSIM_TEST_JOBS"):
        runtime.Builder.from_env({"MADSIM_TEST_JOBS": "65536"})
    # Some(Duration::ZERO) is a limit (panics at the first idle advance), not "no limit"
    assert runtime.Builder(time_limit=0.0).limits().time_limit_ns == 1
    assert runtime.Builder().limits().time_limit_ns == 0


def test_context_api_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(runtime.MadsimHipError, match="no CPU fallback"):
        runtime.Context(0)
    L = runtime.lib()
    assert L.madsim_hip_default_ctx() is None
    cfg, lim, summ = A.Config.default(), A.Limits(), A.Summary()
    w = workload.pingpong(2, 1)
    arr = (C.c_void_p * 1)(None)
    assert L.madsim_hip_run_batch_multi(arr, 1, w.ref(), C.byref(cfg), 0, 0, C.byref(lim), None, C.byref(summ), 1) == -3
    rep = A.Campaign()
    assert L.madsim_hip_run_campaign_multi(arr, 1, w.ref(), C.byref(cfg), 0, 100, 0, 0, 0, C.byref(lim), C.byref(rep)) == -3      # null context
    assert L.madsim_hip_run_campaign_multi(arr, 0, w.ref(), C.byref(cfg), 0, 100, 0, 0, 0, C.byref(lim), C.byref(rep)) == -1      # no contexts
    assert L.madsim_hip_run_campaign_multi(arr, 1, w.ref(), C.byref(cfg), 0, 100, 0, 0, 0, C.byref(lim), None) == -1            # no report
