//! `Builder`: madsim's seed driver (`madsim::runtime::Builder`, runtime/builder.rs:7-162) over the GPU batch runner.
//!
//! Same public fields, same `MADSIM_TEST_*` environment variables, same outcome: `run_workload` returns when every seed
//! passes and otherwise prints the reference's reproduction note (runtime/mod.rs:205-210) for the failing seed and panics —
//! so `cargo test` verdicts are those of `Builder::run`.  One documented difference: with `jobs > 1` the reference reports
//! the first failing seed to *complete*; a batch has no completion order, so the numerically smallest failing seed is reported.
use crate::workload::Workload;
use madsim_hip_sys as sys;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::sync::OnceLock;
use std::time::{Duration, SystemTime};

/// `madsim::Config.net` (net/network.rs:66-89) + the buggify switch.
#[derive(Clone, Debug)]
pub struct NetConfig {
    pub packet_loss_rate: f64,
    pub send_latency: std::ops::Range<Duration>,
    pub buggify: bool,
    /// Ranges a workload's `set_latency(i)` switches to: `NetSim::update_config(|c| c.send_latency = latency_table[i])`
    /// (`net/mod.rs:138-141`); at most four.
    pub latency_table: Vec<std::ops::Range<Duration>>,
}

impl Default for NetConfig {
    fn default() -> Self {
        NetConfig { packet_loss_rate: 0.0, send_latency: Duration::from_millis(1)..Duration::from_millis(10), buggify: false, latency_table: Vec::new() }
    }
}

impl NetConfig {
    pub fn raw(&self) -> sys::madsim_config_t {
        assert!(self.latency_table.len() <= 4, "at most four latency_table entries");
        let (mut lo, mut hi) = ([0u64; 4], [0u64; 4]);
        for (i, r) in self.latency_table.iter().enumerate() { lo[i] = r.start.as_nanos() as u64; hi[i] = r.end.as_nanos() as u64; }
        sys::madsim_config_t {
            packet_loss_rate: self.packet_loss_rate,
            lat_lo_ns: self.send_latency.start.as_nanos() as u64,
            lat_hi_ns: self.send_latency.end.as_nanos() as u64,
            buggify: self.buggify as u32,
            n_loss_table: 0,
            loss_table: [0.0; 4],
            n_lat_table: self.latency_table.len() as u32,
            reserved0: 0,
            lat_table_lo_ns: lo,
            lat_table_hi_ns: hi,
        }
    }
}

/// A library error (never a test verdict): no GPU, malformed workload, limits that do not fit the device, a runner limit
/// (device capacity / step cap) that persists after the re-runs.
#[derive(Debug)]
pub struct RunError {
    pub code: c_int,
    pub message: String,
}

impl std::fmt::Display for RunError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "madsim_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for RunError {}

fn last_error(code: c_int) -> RunError {
    let (what, detail) = unsafe {
        (CStr::from_ptr(sys::madsim_hip_strerror(code)).to_string_lossy().into_owned(),
         CStr::from_ptr(sys::madsim_hip_last_error()).to_string_lossy().into_owned())
    };
    RunError { code, message: format!("{what}: {detail}") }
}

struct Contexts(Vec<*mut sys::madsim_hip_ctx_t>);
// the library serialises calls per context; the pointers are only handed back to it
unsafe impl Send for Contexts {}
unsafe impl Sync for Contexts {}

/// One context per visible GPU, created once per process (MADSIM_HIP_DEVICES = how many to use; default: all).
fn contexts() -> Result<&'static Contexts, RunError> {
    static CTX: OnceLock<Result<Contexts, (c_int, String)>> = OnceLock::new();
    let r = CTX.get_or_init(|| {
        let want: usize = std::env::var("MADSIM_HIP_DEVICES").ok().and_then(|s| s.parse().ok()).unwrap_or(usize::MAX);
        let mut v = Vec::new();
        while v.len() < want {
            let mut c: *mut sys::madsim_hip_ctx_t = std::ptr::null_mut();
            let rc = unsafe { sys::madsim_hip_ctx_create(v.len() as c_int, &mut c) };
            if rc != 0 {
                if v.is_empty() {
                    let e = last_error(rc);
                    return Err((e.code, e.message));
                }
                break; // device index past the last GPU
            }
            v.push(c);
        }
        Ok(Contexts(v))
    });
    match r {
        Ok(c) => Ok(c),
        Err((code, message)) => Err(RunError { code: *code, message: message.clone() }),
    }
}

/// What the panic of a failing seed says (the reference's messages: task/mod.rs:250,253-258).
pub fn verdict_message(verdict: u32) -> &'static str {
    match verdict {
        sys::MADSIM_PANIC => "a task panicked",
        sys::MADSIM_DEADLOCK => "no events, all tasks will block forever",
        sys::MADSIM_TIME_LIMIT => "time limit exceeded",
        sys::MADSIM_OVERFLOW => "device capacity exceeded (runner limit, not a test verdict)",
        sys::MADSIM_STEP_LIMIT => "step cap reached (runner limit, not a test verdict)",
        sys::MADSIM_UNSUPPORTED => "the seed left the workload model (runner verdict, not a test verdict)",
        sys::MADSIM_INTERNAL => "internal invariant of the device code broke (runner verdict, not a test verdict)",
        _ => "pass",
    }
}

/// runtime/mod.rs:205-210
fn note_seed(seed: u64) {
    eprintln!("note: run with `MADSIM_TEST_SEED={seed}` environment variable to reproduce this error");
}

/// Builds the many-seed run with custom configuration values (the fields of `madsim::runtime::Builder`).
pub struct Builder {
    /// The random seed for test.
    pub seed: u64,
    /// The number of tests.
    pub count: u64,
    /// The number of jobs to run simultaneously (no meaning for a batch: every seed is in flight).
    pub jobs: u16,
    /// The configuration.
    pub config: NetConfig,
    /// The time limit for the test.
    pub time_limit: Option<Duration>,
    /// Enable determinism check.
    pub check: bool,
    /// Allow spawning system thread (accepted, no effect: a GPU lane has no system threads).
    pub allow_system_thread: bool,
    /// Device capacities to start from (no reference counterpart; all zero = defaults).  `limits.no_trace_hash = 1` drops the
    /// determinism-log fingerprint from the results (the reference logs only under `check`, rand.rs:67): 4 % faster on ping-pong.
    pub limits: sys::madsim_limits_t,
}

fn zero_limits() -> sys::madsim_limits_t {
    // plain-old-data: all zero = "pick defaults"
    unsafe { std::mem::zeroed() }
}

impl Builder {
    /// builder.rs:64-118: `MADSIM_TEST_SEED`, `MADSIM_TEST_NUM`, `MADSIM_TEST_JOBS`, `MADSIM_TEST_TIME_LIMIT`,
    /// `MADSIM_TEST_CHECK_DETERMINISM`, `MADSIM_ALLOW_SYSTEM_THREAD` (`MADSIM_TEST_CONFIG` is read by the caller: the TOML
    /// parser lives in madsim).
    pub fn from_env() -> Self {
        let seed: u64 = if let Ok(s) = std::env::var("MADSIM_TEST_SEED") {
            s.parse().expect("MADSIM_TEST_SEED should be an integer")
        } else {
            SystemTime::now().duration_since(SystemTime::UNIX_EPOCH).unwrap().as_nanos() as _
        };
        let jobs: u16 = if let Ok(s) = std::env::var("MADSIM_TEST_JOBS") {
            s.parse().expect("MADSIM_TEST_JOBS should be an integer")
        } else {
            1
        };
        let mut count: u64 = if let Ok(s) = std::env::var("MADSIM_TEST_NUM") {
            s.parse().expect("MADSIM_TEST_NUM should be an integer")
        } else {
            1
        };
        let time_limit = std::env::var("MADSIM_TEST_TIME_LIMIT")
            .ok()
            .map(|s| Duration::from_secs_f64(s.parse::<f64>().expect("MADSIM_TEST_TIME_LIMIT should be an number")));
        let check = std::env::var("MADSIM_TEST_CHECK_DETERMINISM").is_ok();
        if check {
            count = count.max(2);
        }
        let allow_system_thread = std::env::var("MADSIM_ALLOW_SYSTEM_THREAD").is_ok();
        Builder { seed, count, jobs, config: NetConfig::default(), time_limit, check, allow_system_thread, limits: zero_limits() }
    }

    fn raw_limits(&self, with_time_limit: bool) -> sys::madsim_limits_t {
        let mut lim = self.limits;
        if with_time_limit {
            if let Some(d) = self.time_limit {
                // 0 means None in the C-ABI; Some(Duration::ZERO) panics at the first idle advance (task/mod.rs:253-258), as 1 ns does
                lim.time_limit_ns = (d.as_nanos() as u64).max(1);
            }
        }
        lim
    }

    /// The raw determinism log of one seed (rand.rs:64-88) and its result.
    fn trace(&self, w: &sys::madsim_workload_t, cfg: &sys::madsim_config_t, lim: &sys::madsim_limits_t) -> Result<(Vec<u8>, sys::madsim_result_t), RunError> {
        let ctx = contexts()?.0[0];
        let mut log = vec![0u8; 1 << 20];
        let mut res: sys::madsim_result_t = unsafe { std::mem::zeroed() };
        let n = unsafe { sys::madsim_hip_ctx_trace_seed(ctx, w, cfg, self.seed, lim, log.as_mut_ptr(), log.len() as u64, &mut res) };
        if n < 0 {
            return Err(last_error(n as c_int));
        }
        log.truncate((n as usize).min(log.len()));
        Ok((log, res))
    }

    /// Seed search: seeds `self.seed .. self.seed + self.count` as batches the LIBRARY keeps in flight on its own streams
    /// (`madsim_hip_run_campaign`), stopping at the first batch that holds a failing seed.  Returns the campaign report — the
    /// smallest failing seed, how many seeds were searched — without per-seed results: re-run the reported seed for details
    /// (`MADSIM_TEST_SEED=<seed>`).  This is the "first failing seed per hour" use of `MADSIM_TEST_NUM` at its full rate: one
    /// 65 536-seed batch alone leaves two thirds of the GPU's issue slots idle.
    pub fn search_first_failure(&self, workload: &Workload) -> Result<sys::madsim_campaign_t, RunError> {
        let w = workload.raw();
        let cfg = self.config.raw();
        let lim = self.raw_limits(true);
        let ctx = contexts()?.0[0];
        let mut rep: sys::madsim_campaign_t = unsafe { std::mem::zeroed() };
        let rc = unsafe {
            sys::madsim_hip_ctx_run_campaign(ctx, &w, &cfg, self.seed, self.count, 0, 0, sys::MADSIM_CAMPAIGN_STOP_AT_FAILURE, &lim, &mut rep)
        };
        if rc != 0 {
            return Err(last_error(rc));
        }
        if rep.first_failing_seed != u64::MAX {
            note_seed(rep.first_failing_seed);
        }
        Ok(rep)
    }

    /// Same contract as `Builder::run` (builder.rs:121-162) for a test body registered as a workload: returns the per-seed
    /// results when every seed passes, panics (after the reproduction note) on the smallest failing seed.  Library errors
    /// and runner limits that survive the re-runs come back as `Err`, never as a test failure.
    pub fn run_workload(&self, workload: &Workload) -> Result<Vec<sys::madsim_result_t>, RunError> {
        let w = workload.raw();
        let cfg = self.config.raw();
        if self.check {
            // Runtime::check_determinism (runtime/mod.rs:178-202): run the seed twice, compare the RNG log; no time limit there
            let lim = self.raw_limits(false);
            let (l1, r1) = self.trace(&w, &cfg, &lim)?;
            if r1.verdict >= sys::MADSIM_OVERFLOW {
                return Err(RunError { code: sys::MADSIM_E_LIMITS, message: format!("seed {}: {}", self.seed, verdict_message(r1.verdict)) });
            }
            let (l2, r2) = self.trace(&w, &cfg, &lim)?;
            if l1 != l2 || r1.trace_hash != r2.trace_hash || r1.obs_hash != r2.obs_hash {
                note_seed(self.seed);
                panic!("non-determinism detected");
            }
            if r1.verdict != sys::MADSIM_PASS {
                note_seed(self.seed);
                panic!("{}", verdict_message(r1.verdict));
            }
            return Ok(vec![r1]);
        }
        let lim = self.raw_limits(true);
        let ctxs = contexts()?;
        let mut out: Vec<sys::madsim_result_t> = vec![unsafe { std::mem::zeroed() }; self.count as usize];
        let mut summary: sys::madsim_summary_t = unsafe { std::mem::zeroed() };
        // Builder::run drives every seed from this process (builder.rs:129-150): the batch is sharded over all GPUs from this
        // thread; seeds that outgrow a device capacity or the step cap are re-run inside, compacted into one launch per round.
        let rc = unsafe {
            sys::madsim_hip_run_batch_multi(ctxs.0.as_ptr(), ctxs.0.len() as c_int, &w, &cfg, self.seed, self.count, &lim,
                                            out.as_mut_ptr(), &mut summary, 6)
        };
        if rc != 0 {
            return Err(last_error(rc));
        }
        if summary.n_failed > 0 {
            let runner = |v: u32| v >= sys::MADSIM_OVERFLOW;
            // a genuine test failure wins over unresolved runner limits: its seed and repro note are never hidden
            if let Some(i) = out.iter().position(|r| r.verdict != sys::MADSIM_PASS && !runner(r.verdict)) {
                note_seed(self.seed + i as u64);
                panic!("{}", verdict_message(out[i].verdict));
            }
            let i = out.iter().position(|r| runner(r.verdict)).unwrap();
            return Err(RunError {
                code: sys::MADSIM_E_LIMITS,
                message: format!("seed {}: {} persists after re-runs with larger limits", self.seed + i as u64, verdict_message(out[i].verdict)),
            });
        }
        Ok(out)
    }
}
