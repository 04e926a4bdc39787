//! Rust host side of `libmadsim_hip.so`.
//!
//! * [`workload`] — the body of a `#[madsim::test(workload)]` as an actor program: `WorkloadBuilder` / `TaskBuilder` mirror
//!   `include/madsim_hip.hpp` and `madsim_amd/workload.py` method for method (names = the reference API calls they stand for).
//! * [`builder`] — `Builder { seed, count, jobs, config, time_limit, check }` with `from_env()` and `run_workload()`:
//!   the fields, environment variables and failure behaviour of `madsim::runtime::Builder` (runtime/builder.rs:7-162,
//!   runtime/mod.rs:205-210).  `bindings/rust/patches/0001-builder-run_workload.patch` adds the same method to madsim itself.
//! * [`interp`] (feature `madsim`) — runs a workload of base ops on REAL madsim through its public API, which is how
//!   `tools/ref_twin` executes `workload::pingpong` on the reference executor and on the GPU from one source.
pub mod builder;
pub mod workload;

#[cfg(all(feature = "madsim", madsim))]
pub mod interp;

pub use builder::{Builder, NetConfig, RunError};
pub use madsim_hip_sys as sys;
pub use workload::{pingpong, pingpong_twin, Addr, TaskBuilder, TaskId, Workload, WorkloadBuilder};
