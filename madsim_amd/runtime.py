"""Host-side mirror of madsim's seed driver over the C-ABI library.

`Builder` carries the same public fields as madsim::runtime::Builder
(madsim/src/sim/runtime/builder.rs:7-22), `Builder.from_env()` reads the same
environment variables (builder.rs:64-118) and `Builder.run(workload)` has the
same outcome as builder.rs:121-162: it returns normally when every seed passes
and otherwise raises `SimulationFailure` after printing the reference's
reproduction note (runtime/mod.rs:205-210) for the failing seed.

The only execution engine is libmadsim_hip.so (hand-written gfx950 kernels).
There is no CPU fallback: if the library is missing or no GPU is visible this
module raises, loudly.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MADSIM_HIP_LIB", os.path.join(_HERE, "libmadsim_hip.so"))   # override: A/B builds of the same library


class MadsimHipError(RuntimeError):
    pass


class RunnerLimitExceeded(MadsimHipError):
    """A seed still carries a RUNNER verdict (device capacity / step cap) after every re-run round.  Neither exists in
    the reference (unbounded containers, no step cap), so this is NOT a test failure and carries no
    MADSIM_TEST_SEED reproduction note: raise the limits (madsim_limits_t) instead."""

    def __init__(self, seed, verdict, result):
        self.seed, self.verdict, self.result = seed, verdict, result
        super().__init__(f"seed {seed}: {A.VERDICT_NAMES[verdict]} persists after re-runs with larger limits")


class SimulationFailure(AssertionError):
    """A seed failed (the reference panics here; cargo test would report the test as failed)."""

    def __init__(self, seed, verdict, result):
        self.seed, self.verdict, self.result = seed, verdict, result
        super().__init__(f"seed {seed}: {A.VERDICT_NAMES[verdict]}")


_lib = None


def lib():
    """Load the product library. Raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and "MADSIM_HIP_LIB" not in os.environ:
            # source-only checkout: build the gfx950 library in-tree once (hipcc cross-compiles without a GPU)
            import subprocess
            try:
                subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
            except (OSError, subprocess.CalledProcessError):
                pass
        if not os.path.exists(LIB_PATH):
            raise MadsimHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C madsim_amd/csrc). There is no CPU fallback.")
        # The library keeps up to five batches in flight on its own HIP streams and ROCclr maps a process's streams onto GPU_MAX_HW_QUEUES
        # hardware queues (default 4: madsim_hip_run_batch(262 144) 8.3 ms against ~5 ms).  The library itself never touches the
        # environment (include/madsim_hip.h madsim_hip_prefer_hw_queues): this host does, here, before it loads anything that initialises
        # HIP — unless the application set the variable itself or initialised HIP earlier (import torch first, then its setting stands).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        L = C.CDLL(LIB_PATH)
        L.madsim_hip_version.restype = C.c_uint32
        L.madsim_hip_prefer_hw_queues.argtypes = [C.c_int]
        L.madsim_hip_strerror.restype = C.c_char_p
        L.madsim_hip_strerror.argtypes = [C.c_int]
        L.madsim_hip_last_error.restype = C.c_char_p
        L.madsim_hip_init.argtypes = [C.c_int]
        L.madsim_hip_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64,
                                           C.POINTER(A.Limits), C.c_void_p, C.POINTER(A.Summary)]
        L.madsim_hip_run_batch_device.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64,
                                                  C.c_uint64, C.POINTER(A.Limits), C.c_void_p, C.c_void_p,
                                                  C.POINTER(A.Summary)]
        L.madsim_hip_run_batch_async.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64,
                                                 C.POINTER(A.Limits), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.madsim_hip_timing_ms.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.madsim_hip_trace_seed.restype = C.c_int64
        L.madsim_hip_trace_seed.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64,
                                            C.POINTER(A.Limits), C.c_void_p, C.c_uint64, C.POINTER(A.Result)]
        L.madsim_hip_run_batch_auto.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64,
                                                C.POINTER(A.Limits), C.c_void_p, C.POINTER(A.Summary), C.c_int]
        L.madsim_hip_geometry.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Limits), C.POINTER(A.Geometry)]
        L.madsim_workload_pingpong.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(A.Node), C.POINTER(A.Prog),
                                               C.POINTER(A.Sock), C.POINTER(A.Insn), C.c_uint32,
                                               C.POINTER(A.Workload)]
        ctxp = C.c_void_p
        L.madsim_hip_ctx_create.argtypes = [C.c_int, C.POINTER(ctxp)]
        L.madsim_hip_ctx_destroy.argtypes = [ctxp]
        L.madsim_hip_ctx_device.argtypes = [ctxp]
        L.madsim_hip_default_ctx.restype = ctxp
        L.madsim_hip_ctx_run_batch.argtypes = [ctxp] + L.madsim_hip_run_batch.argtypes
        L.madsim_hip_ctx_run_batch_auto.argtypes = [ctxp] + L.madsim_hip_run_batch_auto.argtypes
        L.madsim_hip_ctx_run_batch_device.argtypes = [ctxp] + L.madsim_hip_run_batch_device.argtypes
        L.madsim_hip_ctx_run_batch_async.argtypes = [ctxp] + L.madsim_hip_run_batch_async.argtypes
        L.madsim_hip_ctx_timing_ms.argtypes = [ctxp] + L.madsim_hip_timing_ms.argtypes
        L.madsim_hip_ctx_trace_seed.restype = C.c_int64
        L.madsim_hip_ctx_trace_seed.argtypes = [ctxp] + L.madsim_hip_trace_seed.argtypes
        L.madsim_hip_run_batch_multi.argtypes = [C.POINTER(ctxp), C.c_int, C.POINTER(A.Workload), C.POINTER(A.Config),
                                                 C.c_uint64, C.c_uint64, C.POINTER(A.Limits), C.c_void_p,
                                                 C.POINTER(A.Summary), C.c_int]
        L.madsim_hip_run_campaign.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64, C.c_uint64,
                                              C.c_uint32, C.c_uint32, C.POINTER(A.Limits), C.POINTER(A.Campaign)]
        L.madsim_hip_ctx_run_campaign.argtypes = [ctxp] + L.madsim_hip_run_campaign.argtypes
        L.madsim_hip_run_campaign_multi.argtypes = [C.POINTER(ctxp), C.c_int] + L.madsim_hip_run_campaign.argtypes
        if L.madsim_hip_version() != A.ABI_VERSION:
            raise MadsimHipError("libmadsim_hip.so ABI version mismatch")
        # build identity: MADSIM_HIP_LIB may name an A/B build of THIS library (tools/build_variant.sh), nothing else — an
        # object that merely exports the same symbols (the host-compiled test harness, a stub) is refused
        try:
            L.madsim_hip_build_info.restype = C.c_char_p
            info = L.madsim_hip_build_info().decode()
        except AttributeError:
            raise MadsimHipError(f"{LIB_PATH} exports no madsim_hip_build_info(): not a libmadsim_hip.so of this ABI")
        fields = dict(f.split("=", 1) for f in info.split() if "=" in f)
        if not info.startswith("madsim_hip ") or fields.get("arch") != "gfx950" or fields.get("abi") != f"{A.ABI_VERSION}u" \
                or fields.get("backend") != "hip-rocm":
            raise MadsimHipError(f"{LIB_PATH} is not the gfx950 HIP build of ABI {A.ABI_VERSION}: build info {info!r}")
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        L = lib()
        raise MadsimHipError(f"{L.madsim_hip_strerror(rc).decode()}: {L.madsim_hip_last_error().decode()}")
    return rc


_inited_device = None


def init(device=0):
    """Bind this process to one GPU (one process per GPU)."""
    global _inited_device
    _check(lib().madsim_hip_init(device))
    _inited_device = device


def shutdown():
    global _inited_device
    if _lib is not None:
        _lib.madsim_hip_shutdown()
    _inited_device = None


def run_batch(workload, seed0, count, config=None, limits=None):
    """Host-buffer entry point: returns (results ndarray[RESULT_DTYPE], Summary)."""
    if _inited_device is None:
        init(0)
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    out = np.zeros(count, dtype=A.RESULT_DTYPE)
    summ = A.Summary()
    _check(lib().madsim_hip_run_batch(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                      out.ctypes.data_as(C.c_void_p), C.byref(summ)))
    return out, summ


def grow_limits(lim):
    """Double every device capacity (used to re-run seeds that came back MADSIM_OVERFLOW)."""
    g = A.Limits()
    g.time_limit_ns, g.max_steps, g.lanes_per_wave = lim.time_limit_ns, lim.max_steps, 0
    g.heap_lds_slots = lim.heap_lds_slots or 8
    g.heap_spill_slots = max(64, 2 * (lim.heap_spill_slots or 56))
    g.max_tasks = min(254, 2 * (lim.max_tasks or 16))
    g.mbox_regs = min(255, 2 * (lim.mbox_regs or 2))
    g.mbox_msgs = min(255, 2 * (lim.mbox_msgs or 2))
    g.max_conns = min(127, 2 * (lim.max_conns or 4))
    g.chan_queue = min(15, 2 * (lim.chan_queue or 2))
    return g


def run_batch_auto(workload, seed0, count, config=None, limits=None, max_rounds=5):
    """madsim_hip_run_batch_auto: run_batch, then the seeds that came back with a runner verdict (a device capacity, the
    step cap) are run again — all of them in one compacted launch per round — with doubled capacities / a 16x step cap
    (the reference's containers are unbounded and it has no step cap; neither verdict is ever a final answer)."""
    if _inited_device is None:
        init(0)
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    out = np.zeros(count, dtype=A.RESULT_DTYPE)
    summ = A.Summary()
    _check(lib().madsim_hip_run_batch_auto(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                           out.ctypes.data_as(C.c_void_p), C.byref(summ), max_rounds))
    return out, summ


def run_campaign(workload, seed0, total, batch=0, in_flight=0, stop_at_failure=False, config=None, limits=None):
    """madsim_hip_run_campaign: `total` seeds as batches kept in flight on the library's own streams; returns the Campaign
    report (first failing seed, counts) — no per-seed results.  stop_at_failure: stop launching once a completed batch holds a
    seed with a genuine verdict."""
    if _inited_device is None:
        init(0)
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    rep = A.Campaign()
    _check(lib().madsim_hip_run_campaign(workload.ref(), C.byref(cfg), seed0, total, batch, in_flight,
                                         A.CAMPAIGN_STOP_AT_FAILURE if stop_at_failure else 0, C.byref(lim), C.byref(rep)))
    return rep


def run_batch_device(workload, seed0, count, d_out_ptr, stream_ptr=0, config=None, limits=None, want_summary=True):
    """Device-resident entry point: results stay in HBM at `d_out_ptr` (48 B/seed)."""
    if _inited_device is None:
        init(0)
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    summ = A.Summary()
    _check(lib().madsim_hip_run_batch_device(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                             C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr),
                                             C.byref(summ) if want_summary else None))
    return summ


def run_batch_async(workload, seed0, count, d_out_ptr, d_summary_ptr=0, stream_ptr=0, config=None, limits=None,
                    timing_slot=-1):
    """Queue one batch on `stream_ptr` without any host synchronisation (results and the 4-word summary stay in HBM)."""
    if _inited_device is None:
        init(0)
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    _check(lib().madsim_hip_run_batch_async(workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                            C.c_void_p(d_out_ptr), C.c_void_p(d_summary_ptr), C.c_void_p(stream_ptr),
                                            timing_slot))


class Context:
    """madsim_hip_ctx_t: the runner's state on ONE GPU.  A process may hold several (one per GPU); calls on one
    context are serialised by the library, distinct contexts share nothing."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().madsim_hip_ctx_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().madsim_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def run_batch(self, workload, seed0, count, config=None, limits=None, auto_rounds=0):
        cfg, lim = config or A.Config.default(), limits or A.Limits()
        out = np.zeros(count, dtype=A.RESULT_DTYPE)
        summ = A.Summary()
        if auto_rounds:
            _check(lib().madsim_hip_ctx_run_batch_auto(self._h, workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                                       out.ctypes.data_as(C.c_void_p), C.byref(summ), auto_rounds))
        else:
            _check(lib().madsim_hip_ctx_run_batch(self._h, workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                                  out.ctypes.data_as(C.c_void_p), C.byref(summ)))
        return out, summ


def run_batch_multi(contexts, workload, seed0, count, config=None, limits=None, max_rounds=5):
    """madsim_hip_run_batch_multi: one process, one host thread, the seed range sharded contiguously over `contexts`
    (all devices' kernels in flight together), runner verdicts re-run compacted, reports folded on the host."""
    cfg, lim = config or A.Config.default(), limits or A.Limits()
    out = np.zeros(count, dtype=A.RESULT_DTYPE)
    summ = A.Summary()
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _check(lib().madsim_hip_run_batch_multi(arr, len(contexts), workload.ref(), C.byref(cfg), seed0, count, C.byref(lim),
                                            out.ctypes.data_as(C.c_void_p), C.byref(summ), max_rounds))
    return out, summ


def run_campaign_multi(contexts, workload, seed0, total, batch=0, in_flight=0, stop_at_failure=False, config=None, limits=None):
    """madsim_hip_run_campaign_multi: the seed search over several contexts (one per GPU) from one host thread — batch k on context
    k % n, reports read in batch order, every device stopped within one round of batches of the first genuine failure."""
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    rep = A.Campaign()
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _check(lib().madsim_hip_run_campaign_multi(arr, len(contexts), workload.ref(), C.byref(cfg), seed0, total, batch, in_flight,
                                               A.CAMPAIGN_STOP_AT_FAILURE if stop_at_failure else 0, C.byref(lim), C.byref(rep)))
    return rep


def run_campaign_over_ranks(workload, seed0, total, batch=65536, stop_at_failure=True, config=None, limits=None, device_tensors=None, group=None,
                            context=None, in_flight=0, round_batches=0):
    """The seed search with ONE PROCESS PER GPU (torch.distributed; RCCL when the backend is "nccl"): the range is cut into chunks of
    `round_batches` batches (0 = four times the batches the library keeps in flight: rounds long enough for the pipeline to reach its
    steady rate, short enough that an early stop wastes little), chunk c belongs to rank c % world, every rank runs its chunk as ONE
    `madsim_hip_run_campaign` call on its own GPU — batches in flight on the library's streams, stopping inside the chunk at a genuine
    failure — and one all-gather of the 56-byte reports per round lets every rank fold the same answer (madsim_amd/dist.py
    campaign_over_ranks).  Without a process group: the single-GPU search.

    The rank's GPU is explicit: `context` (a runtime.Context on it), or the process-default context, which must have been bound with
    `runtime.init(local_rank)` — this function never initialises it by itself (every rank would land on GPU 0).  `device_tensors`
    defaults to that GPU under the "nccl" backend and to the CPU otherwise (gloo)."""
    import torch.distributed as tdist
    from madsim_amd import dist as mdist
    if context is None and _inited_device is None:
        raise RuntimeError("run_campaign_over_ranks: bind this rank's GPU first — runtime.init(local_rank) — or pass context=runtime.Context(local_rank)")
    dev_index = context.device if context is not None else _inited_device
    if device_tensors is None:
        device_tensors = "cpu"
        if tdist.is_available() and tdist.is_initialized() and tdist.get_backend(group) == "nccl":
            import torch
            device_tensors = torch.device("cuda", dev_index)
    cfg, lim = config or A.Config.default(), limits or A.Limits()
    if not round_batches:
        g = geometry(workload, lim)
        fl = in_flight or (5 if g.blocks_per_cu * g.block_threads // 64 >= 16 else 4 if (g.variant & 16) and g.heap_spill_slots else 3)
        round_batches = 4 * fl

    def chunk(seed_lo, n):
        rep = A.Campaign()
        flags = A.CAMPAIGN_STOP_AT_FAILURE if stop_at_failure else 0
        if context is not None:
            _check(lib().madsim_hip_ctx_run_campaign(context._h, workload.ref(), C.byref(cfg), seed_lo, n, batch, in_flight, flags, C.byref(lim), C.byref(rep)))
        else:
            _check(lib().madsim_hip_run_campaign(workload.ref(), C.byref(cfg), seed_lo, n, batch, in_flight, flags, C.byref(lim), C.byref(rep)))
        return rep.first_failing_seed, rep.n_failed, rep.n_runner, rep.total_steps, rep.total_clock_ns, rep.seeds_run, rep.batches_run
    return mdist.campaign_over_ranks(chunk, seed0, total, batch, stop_at_failure, device_tensors, group, round_batches)


def timing_ms(slot):
    ms = C.c_double(0.0)
    _check(lib().madsim_hip_timing_ms(slot, C.byref(ms)))
    return ms.value


def trace_seed(workload, seed, config=None, limits=None, cap=1 << 20):
    """Determinism log of one seed as produced on the GPU (rand.rs:64-88 format)."""
    if _inited_device is None:
        init(0)
    cfg = config or A.Config.default()
    lim = limits or A.Limits()
    buf = (C.c_uint8 * cap)()
    res = A.Result()
    n = _check(lib().madsim_hip_trace_seed(workload.ref(), C.byref(cfg), seed, C.byref(lim), buf, cap, C.byref(res)))
    return bytes(buf[:min(n, cap)]), res


def geometry(workload, limits=None):
    lim = limits or A.Limits()
    g = A.Geometry()
    _check(lib().madsim_hip_geometry(workload.ref(), C.byref(lim), C.byref(g)))
    return g


def variant_name(g):
    """The sim_kernel specialisation a madsim_geometry_t selects (madsim_k_launch_sim's dispatch), as rocprofv3 names it."""
    b = lambda x: "true" if x else "false"
    lws = (g.variant >> 16) & 0xf
    return (f"sim_kernel<Variant<false, {b(g.variant & 1)}, {-1 if lws == 15 else lws}, {(g.variant >> 8) & 0xff}, "
            f"{b(g.variant & 4)}, {b(g.variant & 16)}>>")


class Builder:
    """madsim::runtime::Builder (runtime/builder.rs:7-22) over the GPU batch runner."""

    DEFAULT_MAX_STEPS = 1 << 24     # device safety net of the first pass (not a reference concept): seeds that reach it
                                    # are re-run ONCE with a 16x cap (madsim_limits_t.max_steps_ceiling, default 1 << 28;
                                    # Builder.max_steps_ceiling raises it); one that still reaches the cap is reported as
                                    # RunnerLimitExceeded, never as a test failure

    def __init__(self, seed=0, count=1, jobs=1, config=None, time_limit=None, check=False,
                 allow_system_thread=False):
        # Rust's types: seed u64, count u64, jobs u16 (builder.rs:7-22); `seed + i` must not wrap (builder.rs:129)
        if not (0 <= seed <= A.U64_MAX):
            raise ValueError("seed must fit u64")
        if not (0 <= count <= A.U64_MAX) or seed + count > A.U64_MAX + 1:
            raise ValueError("count must fit u64 and seed + count must not exceed 2^64")
        if not (0 <= jobs <= 0xFFFF):
            raise ValueError("jobs must fit u16")
        if time_limit is not None and not (time_limit >= 0):
            raise ValueError("time_limit must be a non-negative number of seconds")
        self.seed, self.count, self.jobs = seed, count, jobs
        self.config = config or A.Config.default()
        self.time_limit = time_limit          # seconds (float) or None
        self.check = check
        self.allow_system_thread = allow_system_thread
        self.max_steps_ceiling = 0            # 0 = the library's default (1 << 28); not a reference field
        self.trace_hash = True                # False: madsim_limits_t.no_trace_hash — results carry no fingerprint of the determinism
                                              # log (the reference computes log bytes only under check_determinism, rand.rs:67)

    @classmethod
    def from_env(cls, env=None):
        """builder.rs:64-118: MADSIM_TEST_{SEED,NUM,JOBS,CONFIG,TIME_LIMIT,CHECK_DETERMINISM}."""
        env = os.environ if env is None else env
        if "MADSIM_TEST_SEED" in env:
            try:
                seed = int(env["MADSIM_TEST_SEED"])
            except ValueError:
                raise ValueError("MADSIM_TEST_SEED should be an integer")
        else:
            seed = time.time_ns() & A.U64_MAX              # builder.rs:70-73: UNIX-epoch nanos as u64
        try:
            jobs = int(env.get("MADSIM_TEST_JOBS", "1"))
        except ValueError:
            raise ValueError("MADSIM_TEST_JOBS should be an integer")
        if not (0 <= seed <= A.U64_MAX):                   # `.parse::<u64>()` (builder.rs:66-69)
            raise ValueError("MADSIM_TEST_SEED should be an integer")
        if not (0 <= jobs <= 0xFFFF):                      # `.parse::<u16>()` (builder.rs:75-78)
            raise ValueError("MADSIM_TEST_JOBS should be an integer")
        config = _parse_config(open(env["MADSIM_TEST_CONFIG"]).read()) if "MADSIM_TEST_CONFIG" in env \
            else A.Config.default()
        try:
            count = int(env.get("MADSIM_TEST_NUM", "1"))
        except ValueError:
            raise ValueError("MADSIM_TEST_NUM should be an integer")
        if not (0 <= count <= A.U64_MAX):
            raise ValueError("MADSIM_TEST_NUM should be an integer")
        time_limit = None
        if "MADSIM_TEST_TIME_LIMIT" in env:
            try:
                time_limit = float(env["MADSIM_TEST_TIME_LIMIT"])
            except ValueError:
                raise ValueError("MADSIM_TEST_TIME_LIMIT should be an number")
        check = "MADSIM_TEST_CHECK_DETERMINISM" in env
        if check:
            count = max(count, 2)
        return cls(seed, count, jobs, config, time_limit, check, "MADSIM_ALLOW_SYSTEM_THREAD" in env)

    def limits(self):
        lim = A.Limits()
        if self.time_limit is not None:
            # time_limit_ns == 0 means None in the C-ABI; Some(Duration::ZERO) panics at the first idle advance
            # (task/mod.rs:253-258: `elapsed >= limit`), which a 1 ns limit reproduces exactly
            lim.time_limit_ns = max(1, int(round(self.time_limit * 1e9)))
        lim.max_steps_ceiling = self.max_steps_ceiling
        lim.no_trace_hash = 0 if self.trace_hash else 1
        return lim

    def run(self, workload):
        """builder.rs:121-162. Returns the result array; raises SimulationFailure on the first failing seed.

        Difference from the reference, documented in DESIGN.md: with jobs > 1 the reference reports the
        first seed to *complete* with a failure; this reports the numerically smallest failing seed.
        """
        if self.check:
            return self.check_determinism(workload)
        out, summ = run_batch_auto(workload, self.seed, self.count, self.config, self.limits())
        if summ.n_failed:
            v = out["verdict"]
            runner = v >= A.OVERFLOW          # the runner's limits, not the test's verdict
            genuine = np.nonzero((v != A.PASS) & ~runner)[0]
            if len(genuine):                                          # a real test failure wins over unresolved runner limits:
                i = int(genuine[0])                                   # the first failing seed and its repro note are never hidden
                seed, r = self.seed + i, out[i]
                panic_with_info(seed)
                n_unres = int(runner.sum())
                if n_unres:
                    sys.stderr.write(f"note: {n_unres} other seed(s) still carry a runner limit verdict (first: seed "
                                     f"{self.seed + int(np.nonzero(runner)[0][0])}); raise madsim_limits_t to resolve them\n")
                raise SimulationFailure(seed, int(r["verdict"]), r)
            i = int(np.nonzero(runner)[0][0])
            raise RunnerLimitExceeded(self.seed + i, int(out[i]["verdict"]), out[i])
        return out

    def check_determinism(self, workload):
        """Runtime::check_determinism (runtime/mod.rs:178-202): run the seed twice, compare the RNG log."""
        # check_determinism builds its Runtimes without a time limit (runtime/mod.rs:178-202 never calls set_time_limit)
        lim = A.Limits()
        for _ in range(4):                                  # a runner verdict says nothing about determinism: grow the limits
            log1, r1 = trace_seed(workload, self.seed, self.config, lim)
            if r1.verdict not in (A.OVERFLOW, A.STEP_LIMIT):      # (UNSUPPORTED / INTERNAL: larger limits change nothing)
                break
            lim = grow_limits(lim)
            lim.max_steps = min((lim.max_steps or self.DEFAULT_MAX_STEPS) * 16, 1 << 28)
        if A.is_runner_verdict(r1.verdict):
            raise RunnerLimitExceeded(self.seed, r1.verdict, r1)
        log2, r2 = trace_seed(workload, self.seed, self.config, lim)
        if log1 != log2 or r1.astuple() != r2.astuple():
            panic_with_info(self.seed)
            raise SimulationFailure(self.seed, A.PANIC, r2)       # "non-determinism detected"
        if r1.verdict != A.PASS:
            panic_with_info(self.seed)
            raise SimulationFailure(self.seed, r1.verdict, r1)
        return r1


def panic_with_info(seed):
    """runtime/mod.rs:205-210"""
    sys.stderr.write(f"note: run with `MADSIM_TEST_SEED={seed}` environment variable to reproduce this error\n")


def _parse_config(text):
    """The [net] table of madsim's TOML Config (config.rs:10-43, network.rs:66-89)."""
    try:
        import tomllib as toml            # py311+
    except ImportError:                    # pragma: no cover
        import tomli as toml
    doc = toml.loads(text)
    net = doc.get("net", {})
    cfg = A.Config.default(packet_loss_rate=float(net.get("packet_loss_rate", 0.0)))
    lat = net.get("send_latency")
    if lat:
        cfg.lat_lo_ns = int(lat["start"]["secs"]) * 1_000_000_000 + int(lat["start"]["nanos"])
        cfg.lat_hi_ns = int(lat["end"]["secs"]) * 1_000_000_000 + int(lat["end"]["nanos"])
    return cfg
