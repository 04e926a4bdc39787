"""Geometry <-> kernel-build consistency, on the CPU.

`madsim_hip_geometry` runs the same `make_geometry` + `select_variant` a launch runs (madsim_amd/csrc/geometry.h, sim_kernel.h) and
needs no GPU.  This walks every bench case, every lifecycle workload and a block of every fuzz generator through it for
state_mem in {AUTO, LDS, GLOBAL, COMPACT} x {plain, | DEDUP_TIMERS} x lanes_per_wave in {0, 8, 16, 32, 64} and checks, from the OUTSIDE
(the reported madsim_geometry_t only), the contract `variant_mismatch()` enforces from the inside:

* a combination is either a geometry or MADSIM_E_LIMITS / MADSIM_E_WORKLOAD — never anything else, so never a launch of a build that
  does not fit (round 4 once ran the 64-lane every-class build on a 32-lane geometry: a hung GPU box);
* the build's compile-time lane stride is the geometry's, or the build takes it at run time;
* global-state build <=> a global block per seed; register ready queue => <= 8 tasks, full waves, base ops;
* the build carries every op class the workload uses (recomputed here from the instruction table);
* the workgroups the geometry puts on a CU fit its 160 KiB of LDS (1 280-byte allocation granules), the workgroup is 1/2/4 waves;
* the named build exists (the set of MADSIM_FOR_EACH_VARIANT, restated below and cross-checked against sim_kernel.h's text).
"""
import os
import random
import re

import pytest

from madsim_amd import _abi as A
from madsim_amd import runtime
from madsim_amd import workload as W
from tests import fuzz, lifecycle_workloads as LW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LDS_PER_CU = 160 * 1024
E_WORKLOAD, E_LIMITS = -4, -5          # include/madsim_hip.h
FEAT = dict(TIME=1, CHAN=2, RPC=4, NODE=8, ADDR=16, ALL=31, NOLOG=32, COMPACT=64, NARROW=128)


def compiled_variants():
    """(spill, lws, feat, rq, g) of every non-trace build: parsed from sim_kernel.h's MADSIM_FOR_EACH_VARIANT."""
    text = open(os.path.join(ROOT, "madsim_amd", "csrc", "sim_kernel.h")).read()
    body = text[text.index("#define MADSIM_FOR_EACH_VARIANT(X)"):text.index("#endif", text.index("#define MADSIM_FOR_EACH_VARIANT(X)"))]
    out = set()
    for m in re.finditer(r"X\((true|false), (true|false), (-?\d+), ([^,]+(?:\|[^,]+)*), (true|false), (true|false)\)", body):
        trace, spill, lws, feat, rq, g = m.groups()
        f = eval(feat.replace("MADSIM_FEAT_", ""), {}, dict(FEAT))
        if trace == "false":
            out.add((spill == "true", int(lws), f, rq == "true", g == "true"))
    return out


COMPILED = compiled_variants()


def decode(v):
    """madsim_geometry_t.variant -> (spill, lws, feat, rq, g)  (include/madsim_hip.h)."""
    lws = (v >> 16) & 0xf
    return (bool(v & 1), -1 if (v & 8) else lws, (v >> 8) & 0xff, bool(v & 4), bool(v & 16))


def classes_needed(w):
    """The op classes a workload's instruction table asks for, as geometry.h derives them (a superset check: ADDR and the
    spawn-in-Drop rule are left to the library — the build may carry MORE classes than this, never fewer)."""
    s = w.struct
    ops = {w.insns[i].op for i in range(s.n_insns)}
    O = A.OP
    f = 0
    if ops & {O["RECV_TIMEOUT"], O["MARK"], O["SLEEP_UNTIL"], O["ASSERT_ELAPSED"], O["ADVANCE"], O["TRACE_TIME"]}:
        f |= FEAT["TIME"]
    if ops & {O["CONNECT"], O["ACCEPT"], O["CSEND"], O["CRECV"], O["CCLOSE"]}:
        f |= FEAT["CHAN"]
    if ops & {O["RPC_CALL"], O["RPC_REPLY"]}:
        f |= FEAT["RPC"] | FEAT["TIME"]
    if ops & {O["KILL"], O["RESTART"], O["PAUSE"], O["RESUME"], O["ABORT"], O["ASSERT_EXIT"], O["BUILD"]}:
        f |= FEAT["NODE"]
    return f


def check(w, lim, what):
    L = runtime.lib()
    g = A.Geometry()
    import ctypes as C
    rc = L.madsim_hip_geometry(w.ref(), C.byref(lim), C.byref(g))
    if rc != 0:
        assert rc in (E_LIMITS, E_WORKLOAD), (what, rc, L.madsim_hip_last_error())
        return None
    spill, lws, feat, rq, glob = decode(g.variant)
    lw = g.lanes_per_wave
    assert lw in (8, 16, 32, 64), (what, lw)
    assert (spill, lws, feat, rq, glob) in COMPILED, (what, "not a compiled build", (spill, lws, feat, rq, glob))
    assert lws == -1 or (1 << lws) == lw, (what, f"build compiled for {1 << lws} seed lanes per wave, geometry has {lw}")
    assert glob == (g.global_bytes_per_seed > 0) or (feat & FEAT["COMPACT"]), (what, "global-state build <=> a global block per seed")
    if glob:
        assert lw == 64 or (lw == 32 and (feat & FEAT["ALL"]) == FEAT["TIME"]), (what, "global-state builds: full waves, or 32 lanes timeout-only")
    if rq:
        assert g.max_tasks <= 8 and lw == 64 and not (feat & FEAT["ALL"]), (what, "register ready queue")
    if not spill:
        assert g.heap_spill_slots == 0, (what, "a build without the spill path on a geometry with spilled levels")
    need = classes_needed(w)
    assert need & ~(feat & FEAT["ALL"]) == 0, (what, f"build classes {feat & 31:#x} lack {need & ~feat:#x}")
    if feat & FEAT["NARROW"]:
        assert glob and (lim.state_mem & A.STATE_NARROW_HEAP), (what, "8-byte heap entries: global-state builds, on request only")
    if feat & FEAT["COMPACT"]:
        assert rq and not spill and lw == 64 and g.max_tasks <= 8 and (lim.state_mem & 0xff) in (A.STATE_AUTO, A.STATE_GLOBAL, A.STATE_COMPACT), what   # (GLOBAL: ignored for base ops = AUTO)
    if lim.lanes_per_wave:
        assert lw == lim.lanes_per_wave, (what, "an explicit lanes_per_wave is honoured or refused, never replaced")
    if (lim.state_mem & 0xff) == A.STATE_GLOBAL:      # (documented: base-op workloads have no global-state build and stay in LDS)
        assert glob == bool(feat & FEAT["ALL"]), (what, "explicit MADSIM_STATE_GLOBAL is honoured for extended-op workloads, or refused")
    if (lim.state_mem & 0xff) == A.STATE_LDS:
        assert not glob and not (feat & FEAT["COMPACT"]), (what, "explicit MADSIM_STATE_LDS is honoured or refused")
    if (lim.state_mem & 0xff) == A.STATE_COMPACT:
        assert feat & FEAT["COMPACT"], (what, "explicit MADSIM_STATE_COMPACT is honoured or refused")
    assert g.block_threads in (64, 128, 256), (what, g.block_threads)
    alloc = (g.lds_bytes_per_block + 1279) // 1280 * 1280
    assert g.lds_bytes_per_block <= LDS_PER_CU and g.blocks_per_cu >= 1, (what, g.lds_bytes_per_block)
    assert g.blocks_per_cu * alloc <= max(LDS_PER_CU, alloc), (what, f"{g.blocks_per_cu} workgroups x {alloc} B of LDS on one CU")
    assert g.blocks_per_cu * (g.block_threads // 64) <= 32, (what, "at most 8 waves per SIMD")
    assert g.heap_lds_slots >= 1 and g.grid_blocks >= 1, what
    return g


def _copy(lim):
    c = A.Limits()
    if lim is not None:
        for f, _ in A.Limits._fields_:
            setattr(c, f, getattr(lim, f))
    return c


def combos(base):
    for sm in (A.STATE_AUTO, A.STATE_LDS, A.STATE_GLOBAL, A.STATE_COMPACT):
        for dd in (0, A.STATE_DEDUP_TIMERS, A.STATE_NARROW_HEAP, A.STATE_DEDUP_TIMERS | A.STATE_NARROW_HEAP):
            for lw in (0, 8, 16, 32, 64):
                for nolog in (0, 1):
                    lim = _copy(base)
                    lim.state_mem, lim.lanes_per_wave, lim.no_trace_hash = sm | dd, lw, nolog
                    yield lim, f"state_mem={sm}{'|DEDUP' if dd & A.STATE_DEDUP_TIMERS else ''}{'|NARROW' if dd & A.STATE_NARROW_HEAP else ''} lanes_per_wave={lw} no_trace_hash={nolog}"


def walk(w, base, name):
    n_ok = n_refused = 0
    for lim, desc in combos(base):
        g = check(w, lim, (name, desc))
        n_ok += g is not None
        n_refused += g is None
    return n_ok, n_refused


def test_compiled_set_is_parsed():
    assert len(COMPILED) >= 20 and (False, 6, FEAT["COMPACT"], True, False) in COMPILED and (True, 5, FEAT["TIME"], False, True) in COMPILED


@pytest.mark.parametrize("name", ["pingpong", "raft", "kv", "timers", "topo"])
def test_bench_cases_every_layout_and_lane_count(name):
    w, lim, _ = W.bench_case(name)
    g = check(w, _copy(lim), (name, "as bench.py runs it"))
    assert g is not None, "the bench configuration itself must be a geometry"
    ok, refused = walk(w, lim, name)
    assert ok > 0
    # the shapes the headline numbers come from
    spill, lws, feat, rq, glob = decode(g.variant)
    if name == "pingpong":
        assert feat & FEAT["COMPACT"] and g.blocks_per_cu * g.block_threads // 64 == 16
    if name == "raft":
        assert glob and g.lanes_per_wave == 64 and (feat & 31) == FEAT["TIME"] and (feat & FEAT["NARROW"])     # (full waves again since round 6: 8-byte heap entries)
    if name in ("kv", "topo"):
        assert glob and g.lanes_per_wave == 64 and bool(feat & FEAT["NARROW"]) == (name == "topo")


def test_32_lane_global_layout_only_for_timeout_only_workloads():
    """The slip of round 4, as a rule: lanes_per_wave = 32 with state_mem = GLOBAL is a geometry for timeout-only workloads and
    MADSIM_E_LIMITS for every other extended class — the every-class 32-lane build does not exist."""
    for name in ("kv", "topo"):
        w, lim, _ = W.bench_case(name)
        for lw in (8, 16, 32):
            l2 = _copy(lim); l2.state_mem, l2.lanes_per_wave = A.STATE_GLOBAL, lw
            assert check(w, l2, (name, lw)) is None
    w, lim, _ = W.bench_case("raft")
    for lw, want in ((32, True), (16, False), (8, False), (64, True)):
        l2 = _copy(lim); l2.state_mem, l2.lanes_per_wave = A.STATE_GLOBAL | A.STATE_DEDUP_TIMERS, lw
        assert (check(w, l2, ("raft", lw)) is not None) == want


@pytest.mark.parametrize("name", sorted(LW.ALL))
def test_lifecycle_workloads_every_layout_and_lane_count(name):
    ok, refused = walk(LW.ALL[name](), LW.limits(name), name)
    assert ok > 0


GENS = [("random_workload", fuzz.generous_limits), ("random_lifecycle_workload", fuzz.generous_limits), ("random_rpc_workload", fuzz.generous_limits),
        ("random_addr_workload", fuzz.generous_limits), ("random_ephemeral_workload", fuzz.generous_limits),
        ("random_channel_workload", fuzz.generous_limits), ("random_guard_workload", fuzz.generous_limits),
        ("random_supervisor_workload", fuzz.mixed_limits), ("random_mixed_workload", fuzz.mixed_limits),
        ("random_ipvs_workload", fuzz.generous_limits), ("random_ipvs_runtime_workload", fuzz.generous_limits),
        ("random_timeout_workload", fuzz.mailbox_limits), ("random_reply_without_receive_workload", fuzz.mailbox_limits),
        ("random_unstructured_workload", fuzz.generous_limits), ("random_unstructured_wide_workload", fuzz.wide_limits),
        ("random_latency_workload", fuzz.mailbox_limits)]


@pytest.mark.parametrize("gen,limits", GENS, ids=[g[0] for g in GENS])
def test_fuzz_block_every_layout_and_lane_count(gen, limits):
    """24 programs of every generator, default capacities and the generator's own, through every combination."""
    ok = 0
    for k in range(24):
        w = getattr(fuzz, gen)(random.Random(660_000 + k))[0]
        for base in (None, limits()):
            a, _ = walk(w, base, (gen, k))
            ok += a
    assert ok > 0
