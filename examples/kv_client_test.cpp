// examples/kv_client_test.cpp — an etcd-client-style test on the C++ host mirror: a server listening on 0.0.0.0:2379,
// clients on ephemeral ports, one reliable connection per request, a handler task per connection.
//
// Rust original (the shape of madsim-etcd-client/src/{kv.rs:37-53, server.rs:34-40} and of every madsim-tonic client):
//
//     #[madsim::test]
//     async fn kv_requests() {
//         let h = Handle::current();
//         let server = h.create_node().ip(ip1).build();
//         server.spawn(async move {
//             let ep = Endpoint::bind("0.0.0.0:2379").await.unwrap();
//             loop {
//                 let (tx, mut rx, _) = ep.accept1().await.unwrap();
//                 spawn(async move {                                   // a handler task per connection
//                     let req = *rx.recv().await.unwrap().downcast::<u32>().unwrap();
//                     assert_eq!(req, 0x11);
//                     SERVED.fetch_add(1, Relaxed);
//                     tx.send(Box::new(0x22u32)).await.unwrap();
//                 });
//             }
//         });
//         let clients: Vec<_> = (2..=3).map(|i| h.create_node().ip(ip(i)).build().spawn(async move {
//             let ep = Endpoint::bind("0.0.0.0:0").await.unwrap();     // an ephemeral port (network.rs:224-236)
//             assert_ne!(ep.local_addr().unwrap().port(), 0);
//             sleep(Duration::from_millis(10)).await;
//             for _ in 0..3 {
//                 let (tx, mut rx) = ep.connect1("10.0.0.1:2379".parse().unwrap()).await.unwrap();
//                 tx.send(Box::new(0x11u32)).await.unwrap();
//                 assert_eq!(*rx.recv().await.unwrap().downcast::<u32>().unwrap(), 0x22);
//             }
//         })).collect();
//         for c in clients { c.await.unwrap(); }
//         assert_eq!(SERVED.load(Relaxed), 6);
//     }
//
// Run:  MADSIM_TEST_NUM=4096 ./kv_client_test
#include <cstdio>

#include "../include/madsim_hip.hpp"

int main() {
    using namespace std::chrono_literals;
    madsim::WorkloadBuilder wl;
    const int ns = wl.create_node();
    const int listen = wl.addr(ns, 2379, MADSIM_ADDR_UNSPECIFIED);      // 0.0.0.0:2379 on the server node
    const int dial = wl.addr(ns, 2379);                                 // 10.0.0.1:2379: what the clients connect to
    madsim::Task& handler = wl.task(ns);
    handler.chan_recv().assert_val(0x11).flag_add(0, 1).chan_send(0x22).done();
    madsim::Task& server = wl.task(ns);
    server.bind(listen);
    const int accept_loop = server.label();
    server.accept1(listen).spawn_move_conn(handler).jmp(accept_loop);
    madsim::Task* clients[2];
    for (int i = 0; i < 2; i++) {
        const int nc = wl.create_node();
        const int ep = wl.addr(nc, 0, MADSIM_ADDR_UNSPECIFIED);         // Endpoint::bind("0.0.0.0:0")
        madsim::Task& c = wl.task(nc);
        c.bind(ep, /*port_to_val=*/true).assert_val(1);                 // the node's first ephemeral port
        c.sleep(10ms).set(0, 3);
        const int loop = c.label();
        c.connect1(ep, dial).assert_val(0).chan_send(0x11).chan_recv().assert_val(0x22).chan_close().djnz(0, loop).done();
        clients[i] = &c;
    }
    madsim::Task& m = wl.main();
    m.spawn(server);
    for (auto* c : clients) m.spawn(*c);
    for (auto* c : clients) m.join(*c);
    m.assert_flag(0, 6).done();

    try {
        auto b = madsim::runtime::Builder::from_env();
        auto out = b.run(wl.build());
        std::printf("test kv_requests ... ok (%zu seeds from %llu)\n", out.size(), (unsigned long long)b.seed);
        return 0;
    } catch (const madsim::SimulationFailure& f) {
        std::fprintf(stderr, "test kv_requests ... FAILED: %s\n", f.what());
        return 101;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
