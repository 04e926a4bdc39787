// k_experiment.h — timing-experiment switches for sim_kernel.  NOT part of the product: only tools/build_variant.sh
// defines MADSIM_EXPERIMENT_BUILD, which makes kernel/k_state.h include this file; the product Makefile refuses any
// EXP_* macro (#error in k_state.h).  Builds made with these switches are not bit-exact and are never loaded by
// madsim_amd/ (their libraries are named libmadsim_hip_<tag>.so and selected explicitly through MADSIM_HIP_LIB).
//
//   -DEXP_ALWAYS_ACCEPT   rejection loops accept their first draw (what would the kernel cost without retries?)
//   -DEXP_NOLOG           no determinism-log byte per GlobalRng::with
//   -DEXP_PROF / EXP_PROF2  s_memtime probes around the phases of an iteration / the timer heap (tools/phase_prof.py)
//   -DEXP_NO_LWS_VARIANTS sub-wave lane strides fall back to the runtime-stride build
#ifndef MADSIM_K_EXPERIMENT_H
#define MADSIM_K_EXPERIMENT_H

#ifdef EXP_ALWAYS_ACCEPT
#define EXP_ACCEPT(x) ((x) && false)
#else
#define EXP_ACCEPT(x) (x)
#endif

#ifdef EXP_NOLOG
#define MADSIM_K_LOG_ENABLED 0
#else
#define MADSIM_K_LOG_ENABLED 1
#endif

#if defined(EXP_PROF) || defined(EXP_PROF2)
#define MADSIM_K_PROF 1
#endif
#ifdef EXP_PROF2
#define PROBE2(i) do { uint64_t t_ = __builtin_readcyclecounter(); L.prof_acc[i] += t_ - L.prof_t; L.prof_t = t_; } while (0)
#else
#define PROBE2(i) do { } while (0)
#endif
#ifdef EXP_PROF
#define PROBE(i) do { uint64_t t_ = __builtin_readcyclecounter(); L.prof_acc[i] += t_ - L.prof_t; L.prof_t = t_; } while (0)
// the top of a poll round, behind timer_flush: bucket 10 = the previous round's [C] tail + the pushes; what PROBE(9) behind the loop then collects is
// the wait for the other lanes' further rounds (the probes are lane 0's: a lane that leaves after one round waits there for the slowest lane)
#define PROBE_FLUSH() PROBE(10)
#else
#define PROBE(i) do { } while (0)
#define PROBE_FLUSH() do { } while (0)
#endif

// -DEXP_HALF_LANES[=n]: only every n-th lane of a wave (default 2) carries seeds, each of them n seeds of the launch one after the other; the
// state layout is packed over the carrying lanes.  What does a wave-iteration cost with half the lanes (fewer divergent paths, half
// the working set), at unchanged LDS per seed?  Results stay complete and bit-exact.
#ifdef EXP_HALF_LANES
#define EXP_LANE_DIV (EXP_HALF_LANES + 0 > 1 ? EXP_HALF_LANES + 0 : 2u)
#else
#define EXP_LANE_DIV 1u
#endif

#ifdef EXP_NO_LWS_VARIANTS
#define MADSIM_K_NO_LWS_VARIANTS 1
#endif

#endif
