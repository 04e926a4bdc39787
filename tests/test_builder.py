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


def test_panic_message_classes():
    """Literal panic messages get the code of their class under the nodes' substring patterns (task/mod.rs:297-300)."""
    wl = W.WorkloadBuilder()
    a = wl.create_node(restart_on_panic_matching=("disk", "net"))
    b = wl.create_node(restart_on_panic_matching=("reset",))
    t = wl.task(a)
    t.panic("disk full"); t.panic("bad disk"); t.panic("network reset"); t.panic("out of memory")
    built = wl.build()
    codes = [built.insns[i].imm for i in range(built.struct.n_insns) if built.insns[i].op == A.OP["PANIC"]]
    assert codes[0] == codes[1]                                  # same class: only node a restarts on them
    assert len({codes[0], codes[2], codes[3]}) == 3
    na, nb = built.nodes[a], built.nodes[b]
    assert sorted(na.match[i] for i in range(na.n_match)) == sorted({codes[0], codes[2]})
    assert [nb.match[i] for i in range(nb.n_match)] == [codes[2]]
    # three classes for one node do not fit its two pattern slots
    wl = W.WorkloadBuilder()
    a = wl.create_node(restart_on_panic_matching=("x",)); b = wl.create_node(restart_on_panic_matching=("xy",)); c = wl.create_node(restart_on_panic_matching=("xyz",))
    t = wl.task(a); t.panic("x"); t.panic("xy"); t.panic("xyz")
    with pytest.raises(ValueError, match="classes"):
        wl.build()
    # numeric and literal forms do not mix in one workload
    wl = W.WorkloadBuilder()
    wl.create_node(restart_on_panic_matching=(1,)); n = wl.create_node(restart_on_panic_matching=("boom",))
    wl.task(n).panic("boom")
    with pytest.raises(ValueError, match="mixed"):
        wl.build()
