//! Raw FFI to `libmadsim_hip.so` — the MI355X (gfx950) many-seed runner behind madsim's `Builder::run`
//! (`madsim/src/sim/runtime/builder.rs:121-162`).  One item per item of `include/madsim_hip.h`, same names, same field
//! order; `tests/test_rust_binding.py` keeps the two in step without a Rust toolchain (field order, widths, constant
//! values, function arity and parameter types).  Regenerate with `python tools/gen_rust_sys.py`.
//!
//! Nothing here has a CPU fallback: without the library or a GPU every entry point returns an error code.
#![allow(non_camel_case_types, non_upper_case_globals)]

use std::os::raw::{c_char, c_int, c_void};

/// Opaque per-GPU runner state (`madsim_hip_ctx_t`).
#[repr(C)]
pub struct madsim_hip_ctx_t {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_insn_t {
    pub op: u8,
    pub a: u8,
    pub b: u16,
    pub imm: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_prog_t {
    pub node: u8,
    pub flags: u8,
    pub entry: u16,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_sock_t {
    pub node: u8,
    pub kind: u8,
    pub port: u16,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_service_t {
    pub vaddr: u8,
    pub n_servers: u8,
    pub servers: [u8; 6],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_node_t {
    pub flags: u8,
    pub n_match: u8,
    pub r#match: [u8; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_workload_t {
    pub n_nodes: u32,
    pub n_progs: u32,
    pub n_socks: u32,
    pub n_insns: u32,
    pub nodes: *const madsim_node_t,
    pub progs: *const madsim_prog_t,
    pub socks: *const madsim_sock_t,
    pub insns: *const madsim_insn_t,
    pub n_services: u32,
    pub panic_dyn_max: u32,
    pub services: *const madsim_service_t,
    pub panic_match: *const u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_config_t {
    pub packet_loss_rate: f64,
    pub lat_lo_ns: u64,
    pub lat_hi_ns: u64,
    pub buggify: u32,
    pub n_loss_table: u32,
    pub loss_table: [f64; 4],
    pub n_lat_table: u32,
    pub reserved0: u32,
    pub lat_table_lo_ns: [u64; 4],
    pub lat_table_hi_ns: [u64; 4],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_limits_t {
    pub time_limit_ns: u64,
    pub max_steps: u32,
    pub heap_lds_slots: u32,
    pub heap_spill_slots: u32,
    pub max_tasks: u32,
    pub mbox_regs: u32,
    pub mbox_msgs: u32,
    pub lanes_per_wave: u32,
    pub max_conns: u32,
    pub chan_queue: u32,
    pub sched: u32,
    pub state_mem: u32,
    pub max_steps_ceiling: u32,
    pub no_trace_hash: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_result_t {
    pub verdict: u32,
    pub steps: u32,
    pub clock_ns: u64,
    pub msg_count: u64,
    pub rng_calls: u64,
    pub trace_hash: u64,
    pub obs_hash: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_summary_t {
    pub first_failing_seed: u64,
    pub n_failed: u64,
    pub total_steps: u64,
    pub total_clock_ns: u64,
    pub kernel_ms: f64,
    pub wall_s: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_campaign_t {
    pub seeds_run: u64,
    pub batches_run: u64,
    pub batches_launched: u64,
    pub first_failing_seed: u64,
    pub n_failed: u64,
    pub n_runner: u64,
    pub total_steps: u64,
    pub total_clock_ns: u64,
    pub kernel_ms: f64,
    pub wall_s: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct madsim_geometry_t {
    pub lds_bytes_per_seed: u32,
    pub lds_bytes_per_block: u32,
    pub block_threads: u32,
    pub blocks_per_cu: u32,
    pub grid_blocks: u32,
    pub heap_lds_slots: u32,
    pub heap_spill_slots: u32,
    pub max_tasks: u32,
    pub lanes_per_wave: u32,
    pub variant: u32,
    pub global_bytes_per_seed: u32,
}

// ---- enum madsim_op / enum madsim_verdict -------------------------------------------------------------------------
pub const MS_OP_DONE: u8 = 0;
pub const MS_OP_SPAWN: u8 = 1;
pub const MS_OP_JOIN: u8 = 2;
pub const MS_OP_ABORT: u8 = 3;
pub const MS_OP_YIELD: u8 = 4;
pub const MS_OP_PANIC: u8 = 5;
pub const MS_OP_SET: u8 = 6;
pub const MS_OP_DJNZ: u8 = 7;
pub const MS_OP_JMP: u8 = 8;
pub const MS_OP_TRACE: u8 = 9;
pub const MS_OP_BUILD: u8 = 15;
pub const MS_OP_SLEEP: u8 = 10;
pub const MS_OP_MARK: u8 = 11;
pub const MS_OP_SLEEP_UNTIL: u8 = 12;
pub const MS_OP_ASSERT_ELAPSED: u8 = 13;
pub const MS_OP_ADVANCE: u8 = 14;
pub const MS_OP_BIND: u8 = 20;
pub const MS_OP_SEND: u8 = 21;
pub const MS_OP_REPLY: u8 = 22;
pub const MS_OP_RECV: u8 = 23;
pub const MS_OP_ASSERT_VAL: u8 = 24;
pub const MS_OP_RECV_TIMEOUT: u8 = 25;
pub const MS_OP_CLOSE: u8 = 26;
pub const MS_OP_KILL: u8 = 30;
pub const MS_OP_RESTART: u8 = 31;
pub const MS_OP_PAUSE: u8 = 32;
pub const MS_OP_RESUME: u8 = 33;
pub const MS_OP_CLOG_NODE: u8 = 34;
pub const MS_OP_UNCLOG_NODE: u8 = 35;
pub const MS_OP_CLOG_LINK: u8 = 36;
pub const MS_OP_UNCLOG_LINK: u8 = 37;
pub const MS_OP_ASSERT_EXIT: u8 = 38;
pub const MS_OP_SET_LOSS: u8 = 39;
pub const MS_OP_SLEEP_RAND: u8 = 40;
pub const MS_OP_GSET: u8 = 41;
pub const MS_OP_GADD: u8 = 42;
pub const MS_OP_ASSERT_G: u8 = 43;
pub const MS_OP_PANIC_IF_G_LT: u8 = 44;
pub const MS_OP_JEQ: u8 = 45;
pub const MS_OP_CONNECT: u8 = 46;
pub const MS_OP_ACCEPT: u8 = 47;
pub const MS_OP_CSEND: u8 = 48;
pub const MS_OP_CRECV: u8 = 49;
pub const MS_OP_CCLOSE: u8 = 50;
pub const MS_OP_RPC_CALL: u8 = 51;
pub const MS_OP_RPC_REPLY: u8 = 52;
pub const MS_OP_RAND_BOOL: u8 = 53;
pub const MS_OP_RANDOM: u8 = 54;
pub const MS_OP_TRACE_TIME: u8 = 55;
pub const MS_OP_HOOK_REQ: u8 = 56;
pub const MS_OP_HOOK_RSP: u8 = 57;
pub const MS_OP_IPVS: u8 = 58;
pub const MS_OP_SET_LATENCY: u8 = 59;
pub const MADSIM_PASS: u32 = 0;
pub const MADSIM_PANIC: u32 = 1;
pub const MADSIM_DEADLOCK: u32 = 2;
pub const MADSIM_TIME_LIMIT: u32 = 3;
pub const MADSIM_OVERFLOW: u32 = 4;
pub const MADSIM_STEP_LIMIT: u32 = 5;
pub const MADSIM_UNSUPPORTED: u32 = 6;
pub const MADSIM_INTERNAL: u32 = 7;

// ---- #define constants -------------------------------------------------------------------------------------------
pub const MADSIM_HIP_ABI_VERSION: u32 = 4;
pub const MADSIM_IPVS_ADD_SERVICE: u32 = 0;
pub const MADSIM_IPVS_DEL_SERVICE: u32 = 1;
pub const MADSIM_IPVS_ADD_SERVER: u32 = 2;
pub const MADSIM_IPVS_DEL_SERVER: u32 = 3;
pub const MADSIM_TAG_RPC_FIRST: u32 = 0x80;
pub const MADSIM_TAG_RPC_LAST: u32 = 0xFD;
pub const MADSIM_SPAWN_MOVE_CONN: u32 = 2;
pub const MADSIM_SPAWN_MOVE_REQUEST: u32 = 4;
pub const MADSIM_VAL_TIMEOUT: u32 = 0xFFFFFFFF;
pub const MADSIM_VAL_REFUSED: u32 = 0xFFFFFFFE;
pub const MADSIM_VAL_RESET: u32 = 0xFFFFFFFD;
pub const MADSIM_VAL_ADDR_NOT_AVAILABLE: u32 = 0xFFFFFFFC;
pub const MADSIM_VAL_ADDR_IN_USE: u32 = 0xFFFFFFFB;
pub const MADSIM_PROG_INIT: u32 = 1;
pub const MADSIM_PROG_PRE: u32 = 2;
pub const MADSIM_PROG_DROP_SPAWN: u32 = 4;
pub const MADSIM_ADDR_IP: u32 = 0;
pub const MADSIM_ADDR_UNSPECIFIED: u32 = 1;
pub const MADSIM_ADDR_LOOPBACK: u32 = 2;
pub const MADSIM_ADDR_VIRTUAL: u32 = 3;
pub const MADSIM_MAX_SERVICES: u32 = 8;
pub const MADSIM_SERVICE_ABSENT: u32 = 0x80;
pub const MADSIM_NODE_RESTART_ON_PANIC: u32 = 1;
pub const MADSIM_NODE_RESTART_MATCHING: u32 = 4;
pub const MADSIM_PANIC_CODE_OTHER: u32 = 255;
pub const MADSIM_NODE_NO_IP: u32 = 2;
pub const MADSIM_LIMIT_NONE: u32 = 0xffffffff;
pub const MADSIM_STATE_AUTO: u32 = 0;
pub const MADSIM_STATE_LDS: u32 = 1;
pub const MADSIM_STATE_GLOBAL: u32 = 2;
pub const MADSIM_STATE_COMPACT: u32 = 3;
pub const MADSIM_STATE_DEDUP_TIMERS: u32 = 0x100;
pub const MADSIM_STATE_NARROW_HEAP: u32 = 0x200;
pub const MADSIM_SCHED_STATIC: u32 = 0;
pub const MADSIM_SCHED_QUEUE: u32 = 1;
pub const MADSIM_MAX_LIVE_TASKS: u32 = 254;
pub const MADSIM_MAX_MBOX_REGS: u32 = 255;
pub const MADSIM_MAX_CHAN_QUEUE: u32 = 15;
pub const MADSIM_MAX_CONNS: u32 = 127;
pub const MADSIM_MAX_MBOX_MSGS: u32 = 255;
pub const MADSIM_MAX_SOCKET_GUARDS: u32 = 127;
pub const MADSIM_E_ARG: c_int = -1;
pub const MADSIM_E_HIP: c_int = -2;
pub const MADSIM_E_NOINIT: c_int = -3;
pub const MADSIM_E_WORKLOAD: c_int = -4;
pub const MADSIM_E_LIMITS: c_int = -5;
pub const MADSIM_CAMPAIGN_STOP_AT_FAILURE: u32 = 1;

#[link(name = "madsim_hip")]
extern "C" {
    pub fn madsim_hip_version() -> u32;
    pub fn madsim_hip_build_info() -> *const c_char;
    pub fn madsim_hip_strerror(code: c_int) -> *const c_char;
    pub fn madsim_hip_last_error() -> *const c_char;
    pub fn madsim_hip_prefer_hw_queues(n: c_int) -> c_int;
    pub fn madsim_hip_ctx_create(device: c_int, out: *mut *mut madsim_hip_ctx_t) -> c_int;
    pub fn madsim_hip_ctx_destroy(ctx: *mut madsim_hip_ctx_t) -> c_int;
    pub fn madsim_hip_ctx_device(ctx: *const madsim_hip_ctx_t) -> c_int;
    pub fn madsim_hip_default_ctx() -> *mut madsim_hip_ctx_t;
    pub fn madsim_hip_init(device: c_int) -> c_int;
    pub fn madsim_hip_shutdown() -> c_int;
    pub fn madsim_hip_run_batch(w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, out: *mut madsim_result_t, summary: *mut madsim_summary_t) -> c_int;
    pub fn madsim_hip_run_batch_auto(w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, out: *mut madsim_result_t, summary: *mut madsim_summary_t, max_rounds: c_int) -> c_int;
    pub fn madsim_hip_run_batch_device(w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, d_out: *mut c_void, stream: *mut c_void, summary: *mut madsim_summary_t) -> c_int;
    pub fn madsim_hip_run_batch_async(w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, d_out: *mut c_void, d_summary4: *mut c_void, stream: *mut c_void, timing_slot: c_int) -> c_int;
    pub fn madsim_hip_timing_ms(timing_slot: c_int, ms: *mut f64) -> c_int;
    pub fn madsim_hip_trace_seed(w: *const madsim_workload_t, cfg: *const madsim_config_t, seed: u64, lim: *const madsim_limits_t, log: *mut u8, cap: u64, out: *mut madsim_result_t) -> i64;
    pub fn madsim_hip_ctx_run_batch(ctx: *mut madsim_hip_ctx_t, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, out: *mut madsim_result_t, summary: *mut madsim_summary_t) -> c_int;
    pub fn madsim_hip_ctx_run_batch_auto(ctx: *mut madsim_hip_ctx_t, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, out: *mut madsim_result_t, summary: *mut madsim_summary_t, max_rounds: c_int) -> c_int;
    pub fn madsim_hip_ctx_run_batch_device(ctx: *mut madsim_hip_ctx_t, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, d_out: *mut c_void, stream: *mut c_void, summary: *mut madsim_summary_t) -> c_int;
    pub fn madsim_hip_ctx_run_batch_async(ctx: *mut madsim_hip_ctx_t, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, d_out: *mut c_void, d_summary4: *mut c_void, stream: *mut c_void, timing_slot: c_int) -> c_int;
    pub fn madsim_hip_ctx_timing_ms(ctx: *mut madsim_hip_ctx_t, timing_slot: c_int, ms: *mut f64) -> c_int;
    pub fn madsim_hip_ctx_trace_seed(ctx: *mut madsim_hip_ctx_t, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed: u64, lim: *const madsim_limits_t, log: *mut u8, cap: u64, out: *mut madsim_result_t) -> i64;
    pub fn madsim_hip_run_batch_multi(ctxs: *const *mut madsim_hip_ctx_t, n_ctx: c_int, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, count: u64, lim: *const madsim_limits_t, out: *mut madsim_result_t, summary: *mut madsim_summary_t, max_rounds: c_int) -> c_int;
    pub fn madsim_hip_ctx_run_campaign(ctx: *mut madsim_hip_ctx_t, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, total: u64, batch: u64, in_flight: u32, flags: u32, lim: *const madsim_limits_t, out: *mut madsim_campaign_t) -> c_int;
    pub fn madsim_hip_run_campaign(w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, total: u64, batch: u64, in_flight: u32, flags: u32, lim: *const madsim_limits_t, out: *mut madsim_campaign_t) -> c_int;
    pub fn madsim_hip_run_campaign_multi(ctxs: *const *mut madsim_hip_ctx_t, n_ctx: c_int, w: *const madsim_workload_t, cfg: *const madsim_config_t, seed0: u64, total: u64, batch: u64, in_flight: u32, flags: u32, lim: *const madsim_limits_t, out: *mut madsim_campaign_t) -> c_int;
    pub fn madsim_hip_geometry(w: *const madsim_workload_t, lim: *const madsim_limits_t, g: *mut madsim_geometry_t) -> c_int;
    pub fn madsim_hip_debug_counters(out16: *mut u64) -> c_int;
    pub fn madsim_workload_pingpong(n_nodes: u32, rounds: u32, nodes: *mut madsim_node_t, progs: *mut madsim_prog_t, socks: *mut madsim_sock_t, insns: *mut madsim_insn_t, cap_insns: u32, w: *mut madsim_workload_t) -> c_int;
}
