// tests/cpp/hpp_config_check.cpp — madsim::Config::from_toml and Builder::from_env's MADSIM_TEST_CONFIG (builder.rs:81-88,
// config.rs:29-35, 45-72).  Prints one line per case: tests/test_builder.py compares them with the Python mirror's parser.
// No GPU, no library call: header-only logic.
#include "../../include/madsim_hip.hpp"

#include <cstdio>

static void show(const char* name, const madsim::Config& c) {
    std::printf("%s %.17g %llu %llu\n", name, c.packet_loss_rate, (unsigned long long)c.send_latency_start_ns, (unsigned long long)c.send_latency_end_ns);
}

int main(int argc, char** argv) {
    for (int i = 1; i + 1 < argc; i += 2) {
        try {
            std::ifstream f(argv[i + 1]); std::stringstream ss; ss << f.rdbuf();
            show(argv[i], madsim::Config::from_toml(ss.str()));
        } catch (const std::invalid_argument& e) { std::printf("%s ERROR %s\n", argv[i], e.what()); }
    }
    if (std::getenv("MADSIM_TEST_CONFIG")) {
        madsim::runtime::Builder b = madsim::runtime::Builder::from_env();
        show("from_env", b.config);
    }
    return 0;
}
