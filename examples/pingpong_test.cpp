// examples/pingpong_test.cpp — what a `#[madsim::test]` looks like on this runner (C++ host mirror).
//
// Rust original (the shape of madsim/src/sim/net/endpoint.rs:549-584 `connect_send_recv`, R rounds):
//
//     #[madsim::test]
//     async fn ping_pong() {
//         let h = Handle::current();
//         let n1 = h.create_node().ip("10.0.0.1".parse().unwrap()).build();
//         let n2 = h.create_node().ip("10.0.0.2".parse().unwrap()).build();
//         let t1 = n1.spawn(async { let ep = Endpoint::bind("10.0.0.1:1").await.unwrap(); sleep(1 s).await;
//                                   for _ in 0..R { ep.send_to("10.0.0.2:1", 1, b"ping").await.unwrap();
//                                                   let (len, _) = ep.recv_from(1, &mut buf).await.unwrap();
//                                                   assert_eq!(&buf[..len], b"pong"); } });
//         let t2 = n2.spawn(async { let ep = Endpoint::bind("10.0.0.2:1").await.unwrap();
//                                   for _ in 0..R { let (len, from) = ep.recv_from(1, &mut buf).await.unwrap();
//                                                   assert_eq!(&buf[..len], b"ping");
//                                                   ep.send_to(from, 1, b"pong").await.unwrap(); } });
//         t1.await.unwrap(); t2.await.unwrap();
//     }
//
// Run:  MADSIM_TEST_SEED=1 MADSIM_TEST_NUM=65536 ./pingpong_test
#include <cstdio>

#include "../include/madsim_hip.hpp"

int main() {
    using namespace std::chrono_literals;
    constexpr uint32_t PING = 0x676E6970, PONG = 0x676E6F70, R = 64;
    madsim::WorkloadBuilder wl;
    int n1 = wl.create_node(), n2 = wl.create_node();
    int a1 = wl.addr(n1, 1), a2 = wl.addr(n2, 1);
    madsim::Task& t1 = wl.task(n1);
    t1.bind(a1).sleep(1s).set(0, R);
    int top1 = t1.label();
    t1.send_to(a1, a2, 1, PING).recv_from(a1, 1).assert_val(PONG).djnz(0, top1).done();
    madsim::Task& t2 = wl.task(n2);
    t2.bind(a2).set(0, R);
    int top2 = t2.label();
    t2.recv_from(a2, 1).assert_val(PING).reply(a2, 1, PONG).djnz(0, top2).done();
    wl.main().spawn(t1).spawn(t2).join(t1).join(t2).done();

    try {
        auto b = madsim::runtime::Builder::from_env();
        auto out = b.run(wl.build());
        double sim_s = 0;
        for (auto& r : out) sim_s += r.clock_ns * 1e-9;
        std::printf("test ping_pong ... ok (%zu seeds from %llu, %.1f simulated seconds)\n", out.size(),
                    (unsigned long long)b.seed, sim_s);
        return 0;
    } catch (const madsim::SimulationFailure& f) {
        std::fprintf(stderr, "test ping_pong ... FAILED: %s\n", f.what());
        return 101;                                   // cargo test's exit code for a failed test
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
