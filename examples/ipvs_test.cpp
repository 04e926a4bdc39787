// examples/ipvs_test.cpp — the reference's `ipvs_load_balance` (madsim/src/sim/net/tcp/mod.rs:254-315) and a supervised node
// with substring panic patterns, on the C++ host mirror.
//
// Rust original of the first half:
//
//     let ipvs = NetSim::current().global_ipvs();
//     ipvs.add_service(ServiceAddr::Tcp("1.1.1.1:80".into()), Scheduler::RoundRobin);
//     ipvs.add_server(ServiceAddr::Tcp("1.1.1.1:80".into()), "10.0.0.1:1");
//     ipvs.add_server(ServiceAddr::Tcp("1.1.1.1:80".into()), "10.0.0.2:1");
//     node1: listener = TcpListener::bind("0.0.0.0:1"); (stream, _) = listener.accept(); read == b"1"
//     node2: the same, read == b"2"
//     node3: stream1 = TcpStream::connect("1.1.1.1:80")   // go to node1
//            stream2 = TcpStream::connect("1.1.1.1:80")   // go to node2
//            stream1.write_all(b"1"); stream2.write_all(b"2");
//
// and of the second: `create_node().init(|| async { sleep(3 s); panic!("disk full") }).restart_on_panic_matching("disk")`.
// Run:  MADSIM_TEST_NUM=4096 ./ipvs_test
#include <cstdio>

#include "ipvs_workload.hpp"

int main() {
    const madsim::Workload w = ipvs_example_workload();
    try {
        madsim::runtime::Builder b = madsim::runtime::Builder::from_env();
        b.run(w);
        std::printf("test ipvs_load_balance ... ok (%llu seeds from %llu)\n", (unsigned long long)b.count, (unsigned long long)b.seed);
        return 0;
    } catch (const madsim::SimulationFailure& e) {
        std::fprintf(stderr, "test ipvs_load_balance ... FAILED: %s\n", e.what());
        return 101;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
