// examples/rpc_restart_test.cpp — a typed-RPC test with a server restart, on the C++ host mirror.
//
// Rust original (shape of madsim's rpc doc example, net/rpc.rs:1-60, plus the kill/restart loop of
// tonic-example/tests/test.rs:198-201):
//
//     #[derive(Serialize, Deserialize, Request)] #[rtype("u8")] struct Req(u8);
//
//     #[madsim::test]
//     async fn rpc_survives_restart() {
//         let h = Handle::current();
//         let server = h.create_node().ip(ip1).init(|| async {
//             let ep = Arc::new(Endpoint::bind(addr1).await.unwrap());
//             ep.add_rpc_handler(|_: Req| async move { 7 });            // recv_from_raw(R::ID) + spawn per request
//             pending::<()>().await
//         }).build();
//         let client = h.create_node().ip(ip2).build();
//         let f = client.spawn(async move {
//             let ep = Endpoint::bind(addr2).await.unwrap();
//             let (mut ok, mut timed_out) = (0, 0);
//             for _ in 0..12 {
//                 match ep.call_timeout(addr1, Req(9), Duration::from_millis(100)).await {
//                     Ok(v) => { assert_eq!(v, 7); ok += 1 } Err(_) => timed_out += 1 }
//                 sleep(Duration::from_millis(50)).await;
//             }
//             assert!(ok >= 4 && timed_out >= 1);
//         });
//         sleep(300 ms).await; h.kill(server.id()); sleep(400 ms).await; h.restart(server.id());
//         f.await.unwrap();
//     }
//
// Run:  MADSIM_TEST_NUM=4096 ./rpc_restart_test
#include <cstdio>

#include "../include/madsim_hip.hpp"

int main() {
    using namespace std::chrono_literals;
    madsim::WorkloadBuilder wl;
    int ns = wl.create_node(), nc = wl.create_node();
    int asv = wl.addr(ns, 1), acl = wl.addr(nc, 1);
    madsim::Task& handler = wl.task(ns);
    handler.rpc_reply(asv, 7).done();
    madsim::Task& server = wl.task(ns, /*init=*/true, /*before_block_on=*/true);
    server.bind(asv);
    int top = server.label();
    server.rpc_recv(asv, 0).spawn_move_request(handler).jmp(top);
    madsim::Task& client = wl.task(nc);
    client.bind(acl).sleep(10ms).set(0, 12);
    int loop = client.label();
    client.rpc_call(acl, asv, 0, 9, 100ms);
    int ok = client.label() + 3;
    client.jeq(7, ok).flag_add(1, 1).jmp(ok + 1);
    client.flag_add(0, 1);
    client.sleep(50ms).djnz(0, loop).panic_if_flag_lt(0, 4).panic_if_flag_lt(1, 1).done();
    wl.main().spawn(client).sleep(300ms).kill(ns).sleep(400ms).restart(ns).join(client).done();

    try {
        auto b = madsim::runtime::Builder::from_env();
        b.capacities.mbox_regs = 8; b.capacities.mbox_msgs = 4;   // timed-out calls leave dead registrations behind;
                                                                  // seeds that still outgrow them are re-run larger
        auto out = b.run(wl.build());
        std::printf("test rpc_survives_restart ... ok (%zu seeds from %llu)\n", out.size(), (unsigned long long)b.seed);
        return 0;
    } catch (const madsim::SimulationFailure& f) {
        std::fprintf(stderr, "test rpc_survives_restart ... FAILED: %s\n", f.what());
        return 101;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
