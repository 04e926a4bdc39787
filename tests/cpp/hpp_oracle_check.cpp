// Runs workloads built with the C++ host mirror (include/madsim_hip.hpp) through the CPU oracle's C twin of the batch entry
// point (madsim_cpu_run_batch, same signature as madsim_hip_run_batch) — test infrastructure: checks on a GPU-less box that
// the C++ DSL emits valid tables with the intended semantics (IPVS services, substring panic patterns).
#include <cstdio>
#include <vector>

#include "../../examples/ipvs_workload.hpp"

extern "C" int madsim_cpu_run_batch(const madsim_workload_t* w, const madsim_config_t* cfg, uint64_t seed0, uint64_t count,
                                    const madsim_limits_t* lim, madsim_result_t* out, madsim_summary_t* summary);

int main() {
    const madsim::Workload w = ipvs_example_workload();
    madsim_workload_t raw = w.raw();
    madsim_config_t cfg = madsim::Config().raw();
    std::vector<madsim_result_t> out(64);
    madsim_summary_t s{};
    int rc = madsim_cpu_run_batch(&raw, &cfg, 0, out.size(), nullptr, out.data(), &s);
    std::printf("rc %d failed %llu dyn_max %u services %u rows %zu\n", rc, (unsigned long long)s.n_failed, raw.panic_dyn_max, raw.n_services, w.panic_match.size());
    return rc == 0 && s.n_failed == 0 && raw.panic_dyn_max == 253 && raw.n_services == 1 ? 0 : 1;
}
