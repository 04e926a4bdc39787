"""Host-side rules of the workload builders (no GPU, no oracle run needed): interning of byte strings, typed-RPC messages and
panic messages — the places where the VM stands a small integer for something the reference moves around as bytes or text."""
import pytest

from madsim_amd import _abi as A
from madsim_amd import workload as W


def test_payload_interning_equal_bytes_equal_values():
    wl = W.WorkloadBuilder()
    assert wl.payload(b"ping") == int.from_bytes(b"ping", "little")          # four bytes stand for themselves
    a, b, c = wl.payload(b"hello world"), wl.payload(b"hello world"), wl.payload(b"hello worle")
    assert a == b != c and a >= W.PAYLOAD_BASE
    assert wl.received(b"hello world", 5) == (wl.payload(b"hello"), 5)       # recv_from copies min(buf.len(), data.len()) bytes
    assert wl.build().payloads[a - W.PAYLOAD_BASE] == b"hello world"


def test_rpc_message_interning():
    wl = W.WorkloadBuilder()
    x, y = wl.rpc_message(("Echo", "hi")), wl.rpc_message(("Echo", "hi"), b"")
    assert x == y != wl.rpc_message(("Echo", "hi"), b"data") and 0 <= x <= 0xFF
    for i in range(256 - 2):
        wl.rpc_message(i)
    with pytest.raises(ValueError, match="256"):
        wl.rpc_message("one too many")
    assert wl.build().rpc_messages[x] == (("Echo", "hi"), b"")


def test_panic_message_rows():
    """`restart_on_panic_matching.iter().any(|s| error_msg.contains(s))` (task/mod.rs:297-300) evaluated per message code: literal
    messages are interned from 254 down, numbers stand for their decimal text, and each node gets a 256-bit row."""
    wl = W.WorkloadBuilder()
    a = wl.create_node(restart_on_panic_matching=("disk", "net"))
    b = wl.create_node(restart_on_panic_matching=("reset",))
    c = wl.create_node(restart_on_panic_matching=(0, "1"))          # numbers and strings mix: both are substrings
    t = wl.task(a)
    t.panic("disk full"); t.panic("bad disk"); t.panic("network reset"); t.panic("out of memory"); t.panic_with_flag(0)
    built = wl.build()
    codes = [built.insns[i].imm for i in range(built.struct.n_insns) if built.insns[i].op == A.OP["PANIC"]][:4]
    assert len(set(codes)) == 4 and min(codes) == 251 and built.struct.panic_dyn_max == 250

    def bit(n, code):
        return (built.panic_match[8 * n + (code >> 5)] >> (code & 31)) & 1
    assert [bit(a, x) for x in codes] == [1, 1, 1, 0] and [bit(b, x) for x in codes] == [0, 0, 1, 0]
    assert [x for x in range(251) if bit(c, x)] == [x for x in range(251) if "0" in str(x) or "1" in str(x)]      # "10", "21", "100" ...
    assert not any(bit(c, x) for x in codes) and not bit(a, 7) and not bit(a, 255)
    # a numeric message code may not reach into the literal range
    wl = W.WorkloadBuilder()
    n = wl.create_node(restart_on_panic_matching=("boom",))
    t = wl.task(n); t.panic("boom"); t.panic(254)
    with pytest.raises(ValueError, match="numeric message codes"):
        wl.build()
    with pytest.raises(ValueError, match="empty pattern"):
        W.WorkloadBuilder().create_node(restart_on_panic_matching=("",))


def test_ipvs_service_table():
    wl = W.WorkloadBuilder()
    n1, n2 = wl.create_node(), wl.create_node()
    v = wl.virtual_addr(1, 80)
    s = wl.ipvs_service(v, [wl.addr(n1, 1), wl.addr(n2, 1)])
    built = wl.build()
    assert s == 0 and built.struct.n_services == 1 and built.services[0].vaddr == v and built.services[0].n_servers == 2
    assert built.socks[v].kind == A.ADDR_VIRTUAL


def test_ipvs_runtime_calls():
    """MS_OP_IPVS as the builder emits it, a service that is only declared, and what validation refuses (library and oracle alike)."""
    import oracle
    from madsim_amd import runtime
    wl = W.WorkloadBuilder()
    n1 = wl.create_node()
    v, a = wl.virtual_addr(1, 80), wl.addr(n1, 1)
    s = wl.ipvs_service(v, absent=True)
    t = wl.main(); t.ipvs_add_service(s); t.ipvs_add_server(s, a); t.ipvs_del_server(s, a); t.ipvs_del_service(s)
    built = wl.build()
    assert built.services[0].n_servers == A.SERVICE_ABSENT
    ins = [(i.op, i.a, i.b, i.imm) for i in built.insns[:4]]
    assert ins == [(A.OP["IPVS"], A.IPVS_ADD_SERVICE, s, 0), (A.OP["IPVS"], A.IPVS_ADD_SERVER, s, a),
                   (A.OP["IPVS"], A.IPVS_DEL_SERVER, s, a), (A.OP["IPVS"], A.IPVS_DEL_SERVICE, s, 0)]
    runtime.geometry(built)
    with pytest.raises(ValueError, match="declared absent"):
        wl.ipvs_service(wl.virtual_addr(2, 80), [a], absent=True)
    for bad_call in (lambda t: t.ipvs_add_service(3), lambda t: t.ipvs_add_server(0, 99), lambda t: t._emit("IPVS", a=7, b=0)):
        wl = W.WorkloadBuilder(); n1 = wl.create_node()
        wl.ipvs_service(wl.virtual_addr(1, 80), [wl.addr(n1, 1)])
        bad_call(wl.main())
        w = wl.build()
        with pytest.raises(runtime.MadsimHipError, match="ipvs"):
            runtime.geometry(w)
        with pytest.raises(RuntimeError):
            oracle.run_batch(w, 0, 1)


def test_cpp_mirror_builds_the_ipvs_example_for_the_oracle(tmp_path):
    """examples/ipvs_workload.hpp (the C++ DSL: IPVS service, virtual address, substring panic patterns) run through the oracle's
    C twin of the batch entry point — no GPU involved: the table is valid, every seed passes, one literal message interned."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "hpp_oracle_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(root, "tests", "cpp", "hpp_oracle_check.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "oracle"), "-lmadsim_oracle", "-Wl,-rpath," + os.path.join(root, "oracle")])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and "failed 0 dyn_max 253 services 1" in p.stdout, p.stdout + p.stderr


def test_reset_node_drop_order_is_refused_not_guessed():
    """NetSim::reset_node drops a killed node's sockets in the order of a seeded HashMap (network.rs:142-147, rand.rs:176-180):
    observable when two of them hold un-accepted connections.  Oracle and library refuse such workloads alike."""
    import oracle
    from tests import emu

    def build(two_listeners, kill):
        wl = W.WorkloadBuilder()
        n, c = wl.create_node(), wl.create_node()
        a, b, me = wl.addr(n, 1), wl.addr(n, 2), wl.addr(c, 1)
        t1 = wl.task(n); t1.bind(a); t1.accept1(a)
        t2 = wl.task(n); t2.bind(b)
        if two_listeners:
            t2.accept1(b)
        cl = wl.task(c); cl.bind(me); cl.connect1(me, a); cl.connect1(me, b)
        m = wl.main(); m.spawn(t1); m.spawn(t2); m.spawn(cl); m.sleep(ms=50)
        if kill:
            m.kill(n)
        m.sleep(ms=50)
        return wl.build()
    for two, kill, ok in ((True, True, False), (True, False, True), (False, True, True)):
        w = build(two, kill)
        if ok:
            oracle.run_batch(w, 0, 4); emu.run_batch(w, 0, 4)
        else:
            with pytest.raises(Exception):
                oracle.run_batch(w, 0, 4)
            with pytest.raises(RuntimeError, match="reset_node"):
                emu.run_batch(w, 0, 4)


def test_cpp_mirror_parses_madsim_test_config_like_the_python_mirror(tmp_path):
    """MADSIM_TEST_CONFIG (builder.rs:81-88): the C++ mirror's TOML subset against the Python mirror's tomli parse, on the
    reference's own test text (config.rs:52-56), on what `Display` prints (sub-table sections), on defaults, and on junk."""
    import os
    import subprocess
    from madsim_amd import runtime
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = {
        "reference_test": '\n        [net]\n        packet_loss_rate = 0.1\n        send_latency = { start = { secs = 0, nanos = 1000000 }, end = { secs = 0, nanos = 10000000 } }\n        \n        [tcp]\n        ',
        "display_form": '[net]\npacket_loss_rate = 0.25\n\n[net.send_latency.start]\nsecs = 1\nnanos = 500_000_000\n\n[net.send_latency.end]\nsecs = 2\nnanos = 0\n\n[tcp]\n',
        "defaults": "# nothing set\n[tcp]\n",
        "loss_only": "[net]\npacket_loss_rate = 1.0 # everything\n",
        "dotted_keys": 'net.packet_loss_rate = 0.5\nnet.send_latency.start = { secs = 0, nanos = 2_000_000 }\nnet.send_latency.end = { "secs" = 0, nanos = 3000000 }\n',
        "unknown_field": "[net]\npacket_loss = 0.1\n",
        "half_range": "[net]\nsend_latency = { start = { secs = 0, nanos = 1 } }\n",
        "not_a_number": "[net]\npacket_loss_rate = lots\n",
    }
    exe = str(tmp_path / "hpp_config_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(root, "tests", "cpp", "hpp_config_check.cpp"), "-o", exe])
    argv = [exe]
    for name, text in cases.items():
        path = tmp_path / (name + ".toml"); path.write_text(text); argv += [name, str(path)]
    env = dict(os.environ, MADSIM_TEST_CONFIG=str(tmp_path / "display_form.toml"), MADSIM_TEST_SEED="1")
    lines = dict(l.split(" ", 1) for l in subprocess.run(argv, capture_output=True, text=True, env=env, check=True).stdout.splitlines())
    for name, text in cases.items():
        if name in ("unknown_field", "half_range", "not_a_number"):
            assert lines[name].startswith("ERROR failed to parse config file"), (name, lines[name])
            continue
        c = runtime._parse_config(text)
        loss, lo, hi = lines[name].split()
        assert (float(loss), int(lo), int(hi)) == (c.packet_loss_rate, c.lat_lo_ns, c.lat_hi_ns), (name, lines[name])
    assert lines["from_env"] == lines["display_form"] and lines["display_form"].split()[1:] == ["1500000000", "2000000000"]
