"""ctypes mirror of include/madsim_hip.h (the C-ABI drop-in boundary).

Pure data definitions: no device code, no oracle.  Both the product loader
(`madsim_amd.runtime`) and the test-only oracle loader (`oracle/`) feed these
same structs, so a parity test hands identical bytes to both sides.
"""
import ctypes as C

ABI_VERSION = 4
U64_MAX = (1 << 64) - 1
LIMIT_NONE = 0xFFFFFFFF
SCHED_STATIC, SCHED_QUEUE = 0, 1
STATE_AUTO, STATE_LDS, STATE_GLOBAL, STATE_COMPACT = 0, 1, 2, 3
STATE_DEDUP_TIMERS = 0x100     # OR-ed into state_mem: re-registered Sleep timers as counts (include/madsim_hip.h)
STATE_NARROW_HEAP = 0x200      # OR-ed into state_mem: 8-byte timer-heap entries + delivery record pool (global-state builds with a spill region)
VAL_TIMEOUT = 0xFFFFFFFF
VAL_REFUSED = 0xFFFFFFFE
VAL_RESET = 0xFFFFFFFD


class Insn(C.Structure):
    _fields_ = [("op", C.c_uint8), ("a", C.c_uint8), ("b", C.c_uint16), ("imm", C.c_uint32)]


class Prog(C.Structure):
    _fields_ = [("node", C.c_uint8), ("flags", C.c_uint8), ("entry", C.c_uint16)]


ADDR_IP, ADDR_UNSPECIFIED, ADDR_LOOPBACK, ADDR_VIRTUAL = 0, 1, 2, 3
MAX_SERVICES = 8
SERVICE_ABSENT = 0x80       # madsim_service_t.n_servers: the address is declared, the service added by a task (MS_OP_IPVS)
IPVS_ADD_SERVICE, IPVS_DEL_SERVICE, IPVS_ADD_SERVER, IPVS_DEL_SERVER = 0, 1, 2, 3
VAL_ADDR_NOT_AVAILABLE, VAL_ADDR_IN_USE = 0xFFFFFFFC, 0xFFFFFFFB
NODE_NO_IP = 2


class Sock(C.Structure):
    _fields_ = [("node", C.c_uint8), ("kind", C.c_uint8), ("port", C.c_uint16)]


class Node(C.Structure):
    _fields_ = [("flags", C.c_uint8), ("n_match", C.c_uint8), ("match", C.c_uint8 * 2)]


class Service(C.Structure):
    """madsim_service_t: one IPVS virtual service (net/ipvs.rs) — address entry + real servers in add_server order."""
    _fields_ = [("vaddr", C.c_uint8), ("n_servers", C.c_uint8), ("servers", C.c_uint8 * 6)]


class Workload(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("n_progs", C.c_uint32), ("n_socks", C.c_uint32), ("n_insns", C.c_uint32),
        ("nodes", C.POINTER(Node)), ("progs", C.POINTER(Prog)), ("socks", C.POINTER(Sock)),
        ("insns", C.POINTER(Insn)),
        ("n_services", C.c_uint32), ("panic_dyn_max", C.c_uint32), ("services", C.POINTER(Service)),
        ("panic_match", C.POINTER(C.c_uint32)),
    ]


HEADER_STRUCTS = {"madsim_service_t": Service}          # + madsim_campaign_t, registered below its definition


class Config(C.Structure):
    """madsim::Config.net (net/network.rs:66-89) + the buggify switch (rand.rs:113-134)."""
    _fields_ = [
        ("packet_loss_rate", C.c_double), ("lat_lo_ns", C.c_uint64), ("lat_hi_ns", C.c_uint64),
        ("buggify", C.c_uint32), ("n_loss_table", C.c_uint32), ("loss_table", C.c_double * 4),
        ("n_lat_table", C.c_uint32), ("reserved0", C.c_uint32), ("lat_table_lo_ns", C.c_uint64 * 4), ("lat_table_hi_ns", C.c_uint64 * 4),
    ]

    @classmethod
    def default(cls, packet_loss_rate=0.0, lat_lo_ns=1_000_000, lat_hi_ns=10_000_000, buggify=False,
                loss_table=(), lat_table=()):
        """`lat_table`: up to four (lo_ns, hi_ns) ranges that MS_OP_SET_LATENCY (TaskBuilder.set_latency) switches between."""
        c = cls()
        c.n_lat_table = len(lat_table)
        for i, (lo, hi) in enumerate(lat_table):
            c.lat_table_lo_ns[i], c.lat_table_hi_ns[i] = lo, hi
        c.packet_loss_rate = packet_loss_rate
        c.lat_lo_ns, c.lat_hi_ns = lat_lo_ns, lat_hi_ns
        c.buggify = 1 if buggify else 0
        c.n_loss_table = len(loss_table)
        for i, p in enumerate(loss_table):
            c.loss_table[i] = p
        return c


class Limits(C.Structure):
    _fields_ = [
        ("time_limit_ns", C.c_uint64), ("max_steps", C.c_uint32), ("heap_lds_slots", C.c_uint32),
        ("heap_spill_slots", C.c_uint32), ("max_tasks", C.c_uint32), ("mbox_regs", C.c_uint32),
        ("mbox_msgs", C.c_uint32), ("lanes_per_wave", C.c_uint32), ("max_conns", C.c_uint32), ("chan_queue", C.c_uint32),
        ("sched", C.c_uint32), ("state_mem", C.c_uint32), ("max_steps_ceiling", C.c_uint32), ("no_trace_hash", C.c_uint32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("verdict", C.c_uint32), ("steps", C.c_uint32), ("clock_ns", C.c_uint64), ("msg_count", C.c_uint64),
        ("rng_calls", C.c_uint64), ("trace_hash", C.c_uint64), ("obs_hash", C.c_uint64),
    ]

    def astuple(self):
        return (self.verdict, self.steps, self.clock_ns, self.msg_count, self.rng_calls, self.trace_hash,
                self.obs_hash)


class Summary(C.Structure):
    _fields_ = [
        ("first_failing_seed", C.c_uint64), ("n_failed", C.c_uint64), ("total_steps", C.c_uint64),
        ("total_clock_ns", C.c_uint64), ("kernel_ms", C.c_double), ("wall_s", C.c_double),
    ]


class Campaign(C.Structure):
    """madsim_campaign_t: the report of madsim_hip_run_campaign (batches kept in flight by the library)."""
    _fields_ = [
        ("seeds_run", C.c_uint64), ("batches_run", C.c_uint64), ("batches_launched", C.c_uint64),
        ("first_failing_seed", C.c_uint64), ("n_failed", C.c_uint64), ("n_runner", C.c_uint64),
        ("total_steps", C.c_uint64), ("total_clock_ns", C.c_uint64), ("kernel_ms", C.c_double), ("wall_s", C.c_double),
    ]


CAMPAIGN_STOP_AT_FAILURE = 1


class Geometry(C.Structure):
    _fields_ = [
        ("lds_bytes_per_seed", C.c_uint32), ("lds_bytes_per_block", C.c_uint32), ("block_threads", C.c_uint32),
        ("blocks_per_cu", C.c_uint32), ("grid_blocks", C.c_uint32), ("heap_lds_slots", C.c_uint32),
        ("heap_spill_slots", C.c_uint32), ("max_tasks", C.c_uint32), ("lanes_per_wave", C.c_uint32),
        ("variant", C.c_uint32), ("global_bytes_per_seed", C.c_uint32),
    ]


HEADER_STRUCTS["madsim_campaign_t"] = Campaign
assert C.sizeof(Insn) == 8 and C.sizeof(Prog) == 4 and C.sizeof(Sock) == 4 and C.sizeof(Node) == 4
assert C.sizeof(Result) == 48 and C.sizeof(Summary) == 48 and C.sizeof(Limits) == 64 and C.sizeof(Config) == 136

# numpy view of a result array: one record per seed, same layout as madsim_result_t
RESULT_DTYPE = [("verdict", "<u4"), ("steps", "<u4"), ("clock_ns", "<u8"), ("msg_count", "<u8"),
                ("rng_calls", "<u8"), ("trace_hash", "<u8"), ("obs_hash", "<u8")]

PASS, PANIC, DEADLOCK, TIME_LIMIT, OVERFLOW, STEP_LIMIT, UNSUPPORTED, INTERNAL = range(8)
VERDICT_NAMES = ["pass", "panic", "deadlock", "time-limit", "resource-overflow", "step-limit", "outside-the-workload-model", "internal-invariant"]


def is_runner_verdict(v):
    """MADSIM_IS_RUNNER_VERDICT: a statement about this runner (capacity, step cap, model frontier, internal), never a test failure."""
    return v >= OVERFLOW

# enum madsim_op
OP = dict(
    DONE=0, SPAWN=1, JOIN=2, ABORT=3, YIELD=4, PANIC=5, SET=6, DJNZ=7, JMP=8, TRACE=9,
    SLEEP=10, MARK=11, SLEEP_UNTIL=12, ASSERT_ELAPSED=13, ADVANCE=14, BUILD=15,
    BIND=20, SEND=21, REPLY=22, RECV=23, ASSERT_VAL=24, RECV_TIMEOUT=25, CLOSE=26,
    KILL=30, RESTART=31, PAUSE=32, RESUME=33, CLOG_NODE=34, UNCLOG_NODE=35, CLOG_LINK=36,
    UNCLOG_LINK=37, ASSERT_EXIT=38, SET_LOSS=39, SLEEP_RAND=40, GSET=41, GADD=42, ASSERT_G=43, PANIC_IF_G_LT=44, JEQ=45, CONNECT=46, ACCEPT=47, CSEND=48, CRECV=49, CCLOSE=50, RPC_CALL=51, RPC_REPLY=52, RAND_BOOL=53, RANDOM=54, TRACE_TIME=55, HOOK_REQ=56, HOOK_RSP=57, IPVS=58, SET_LATENCY=59,
)
PROG_INIT, PROG_PRE, PROG_DROP_SPAWN = 1, 2, 4
NODE_RESTART_ON_PANIC = 1
NODE_RESTART_MATCHING = 4
