// examples/ipvs_workload.hpp — the workload of examples/ipvs_test.cpp as a function, so that the CPU test suite can run the very
// same table through the oracle's C twin (madsim_cpu_run_batch) without a GPU (tests/test_builder.py).
#ifndef MADSIM_EXAMPLES_IPVS_WORKLOAD_HPP
#define MADSIM_EXAMPLES_IPVS_WORKLOAD_HPP
#include "../include/madsim_hip.hpp"

inline madsim::Workload ipvs_example_workload() {
    using namespace std::chrono_literals;
    madsim::WorkloadBuilder wl;
    const int n1 = wl.create_node(), n2 = wl.create_node(), n3 = wl.create_node();
    const int l1 = wl.addr(n1, 1, MADSIM_ADDR_UNSPECIFIED), l2 = wl.addr(n2, 1, MADSIM_ADDR_UNSPECIFIED);
    const int s1 = wl.addr(n1, 1), s2 = wl.addr(n2, 1);                 // the real servers, as add_server names them
    const int vip = wl.virtual_addr(1, 80);                             // "1.1.1.1:80"
    const int svc = wl.ipvs_service(vip, {s1});                         // add_service + add_server(s1) before the run; s2 joins at run time
    const int c = wl.addr(n3, 0, MADSIM_ADDR_UNSPECIFIED);              // TcpStream::connect binds an ephemeral Endpoint
    const uint32_t one = wl.payload("1"), two = wl.payload("2");
    madsim::Task& f1 = wl.task(n1); f1.bind(l1).accept1(l1).chan_recv().assert_val(one);
    madsim::Task& f2 = wl.task(n2); f2.bind(l2).accept1(l2).chan_recv().assert_val(two);
    madsim::Task& hold = wl.task(n3); hold.chan_send(one).sleep(200ms);              // stream1 lives on in its own task
    madsim::Task& f3 = wl.task(n3);
    f3.sleep(50ms).bind(c);
    f3.connect1(c, vip).assert_val(0).spawn_move_conn(hold);                          // go to node1
    f3.ipvs_add_server(svc, s2);                                                      // NetSim::global_ipvs().add_server(.., "10.0.0.2:1")
    f3.connect1(c, vip).assert_val(0).chan_send(two).sleep(200ms);                    // go to node2 (rr_index 1 of [s1, s2])
    // a supervised node: restarts on "disk full" (pattern "disk"), three times within ten seconds at most
    const int nd = wl.create_node_matching({"disk", "net"});
    madsim::Task& flaky = wl.task(nd, /*init=*/true, /*before_block_on=*/true);
    flaky.flag_add(0, 1).sleep(3s).panic("disk full");
    wl.main().spawn(f1).spawn(f2).spawn(f3).join(f1).join(f2).join(f3).sleep(40s).panic_if_flag_lt(0, 3);
    return wl.build();
}
#endif
