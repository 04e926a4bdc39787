"""The Rust side of the boundary (bindings/rust/) against include/madsim_hip.h — without a Rust toolchain.

`north_star` keeps the host code in Rust; this image has no rustc, so the crates cannot be compiled here.  What CAN drift
silently is the FFI surface, and that is checked textually: every #[repr(C)] struct of madsim-hip-sys/src/lib.rs (field
names, order, widths), every constant and every `extern "C"` declaration (name, arity, parameter and return types) is
parsed out of the Rust source and compared with the header as tests/cheader.py parses it."""
import os
import re

import pytest

from tests import cheader as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "rust")
SYS = open(os.path.join(RS, "madsim-hip-sys", "src", "lib.rs")).read()

C2RUST = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "int": "c_int",
          "double": "f64", "char": "c_char", "void": "c_void"}
RUST_SIZE = {"u8": 1, "u16": 2, "u32": 4, "u64": 8, "i64": 8, "c_int": 4, "f64": 8}


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", SYS, flags=re.S):
        fields = []
        for fm in re.finditer(r"pub (r#)?(\w+): ([^,\n]+),", m.group(2)):
            fields.append((fm.group(2), fm.group(3).strip()))
        out[m.group(1)] = fields
    return out


def c_type_to_rust(ct):
    """'const madsim_workload_t*' -> '*const madsim_workload_t'; 'madsim_hip_ctx_t* const*' -> '*const *mut madsim_hip_ctx_t'."""
    m = re.match(r"^(const\s+)?([\w ]+?)\s*((?:\*\s*(?:const\s*)?)*)$", ct.strip())
    const0, base, stars = bool(m.group(1)), m.group(2).strip(), m.group(3)
    t = C2RUST.get(base, base)
    ptrs = re.findall(r"\*\s*(const)?", stars)
    consts = ([const0] + [bool(c) for c in ptrs[:-1]]) if ptrs else []
    for c in consts:
        t = ("*const " if c else "*mut ") + t
    return t


def test_repr_c_structs_match_header():
    rs, cs = rust_structs(), H.structs()
    assert set(cs) <= set(rs), f"structs of the header without a #[repr(C)] twin: {sorted(set(cs) - set(rs))}"
    for name, cfields in cs.items():
        rfields = rs[name]
        assert [f[0] for f in rfields] == [f[0] for f in cfields], (name, "field names / order")
        for (fname, ctype, arr, ptr), (_, rtype) in zip(cfields, rfields):
            want = C2RUST.get(ctype, ctype)
            if ptr:
                want = "*const " + want
            if arr:
                want = f"[{want}; {arr}]"
            assert rtype == want, (name, fname, rtype, want)
    # sizes follow from identical field lists under #[repr(C)]; spot-check the two the ABI pins
    assert H.layout(cs["madsim_result_t"])[1] == 48 and H.layout(cs["madsim_limits_t"])[1] == 64


def test_extern_c_declarations_match_header():
    block = re.search(r'extern "C" \{(.*?)\n\}', SYS, flags=re.S).group(1)
    rfns = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?: -> ([^;]+))?;", block, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        rfns[m.group(1)] = (m.group(3).strip() if m.group(3) else None, params)
    cfns = H.functions()
    assert set(cfns) == set(rfns), (sorted(set(cfns) - set(rfns)), sorted(set(rfns) - set(cfns)))
    for name, (cret, cparams) in cfns.items():
        rret, rparams = rfns[name]
        assert len(rparams) == len(cparams), (name, "arity")
        assert rret == (None if cret == "void" else c_type_to_rust(cret)), (name, "return type", rret, cret)
        for i, (cp, rp) in enumerate(zip(cparams, rparams)):
            assert rp == c_type_to_rust(cp), (name, i, rp, cp)
    assert '#[link(name = "madsim_hip")]' in SYS


def test_constants_match_header():
    raw = open(H.HEADER_PATH).read()
    rconst = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub const (\w+): \w+ = (-?(?:0x[0-9a-fA-F]+|\d+));", SYS)}
    n = 0
    for m in re.finditer(r"\b(MS_OP_\w+|MADSIM_(?:PASS|PANIC|DEADLOCK|TIME_LIMIT|OVERFLOW|STEP_LIMIT|UNSUPPORTED|INTERNAL))\s*=\s*(\d+)", H.header_text()):
        assert rconst[m.group(1)] == int(m.group(2)), m.group(1)
        n += 1
    assert n >= 57 + 6 - 6 and n == len(re.findall(r"pub const (?:MS_OP_\w+|MADSIM_(?:PASS|PANIC|DEADLOCK|TIME_LIMIT|OVERFLOW|STEP_LIMIT|UNSUPPORTED|INTERNAL)): ", SYS))
    for m in re.finditer(r"#define\s+(MADSIM_\w+)\s+\(?(0x[0-9a-fA-F]+|-?\d+)[uU]?\)?", raw):
        if m.group(1) == "MADSIM_HIP_H":
            continue
        assert rconst[m.group(1)] == int(m.group(2), 0), m.group(1)
    assert rconst["MADSIM_HIP_ABI_VERSION"] == int(re.search(r"MADSIM_HIP_ABI_VERSION (\d+)u", raw).group(1))


def test_generator_reproduces_the_committed_binding():
    """tools/gen_rust_sys.py and the committed lib.rs agree (a hand edit must be mirrored in the generator, or vice versa)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], capture_output=True, text=True, check=True).stdout
    assert out == SYS


def test_dsl_pingpong_is_the_c_and_python_table():
    """`madsim_hip::pingpong` (workload.rs) emits, call for call, what workload.py::pingpong emits: the method bodies are
    parsed for the (op, a, b, imm) they push and replayed here in the order pingpong_with() calls them."""
    from madsim_amd import _abi as A
    from madsim_amd import workload as W
    src = open(os.path.join(RS, "madsim-hip", "src", "workload.rs")).read()
    emits = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (\w+)\(&mut self[^)]*\) -> &mut Self \{[^}]*?self\.emit\(sys::MS_OP_(\w+),", src)}
    for meth, op in {"bind": "BIND", "sleep": "SLEEP", "send_to": "SEND", "recv_from": "RECV", "assert_val": "ASSERT_VAL",
                     "reply": "REPLY", "spawn": "SPAWN", "join": "JOIN", "set": "SET", "djnz": "DJNZ", "done": "DONE"}.items():
        assert emits[meth] == op, (meth, emits.get(meth))
    body = src[src.index("fn pingpong_with"):]
    # pinger: bind, sleep(1 s), repeat{send_to, recv_from, assert_val(PONG)}; ponger: bind, repeat{recv_from, assert_val(PING), reply}
    assert re.search(r"t\.bind\(me\);.*?t\.sleep\(Duration::from_secs\(1\)\);.*?t\.send_to\(me, peer, 1, PING\)\.recv_from\(me, 1\)\.assert_val\(PONG\)", body, flags=re.S)
    assert re.search(r"t\.recv_from\(me, 1\)\.assert_val\(PING\)\.reply\(me, 1, PONG\)", body)
    assert "pub const PING: u32 = 0x676E_6970" in src and "pub const PONG: u32 = 0x676E_6F70" in src
    assert (W.PING, W.PONG) == (0x676E6970, 0x676E6F70)
    w = W.pingpong(4, 64)
    ops = [w.insns[i].op for i in range(w.struct.n_insns)]
    O = A.OP
    assert ops[:9] == [O["SPAWN"]] * 4 + [O["JOIN"]] * 4 + [O["DONE"]]
    assert ops[9:18] == [O["BIND"], O["SLEEP"], O["SET"], O["SEND"], O["RECV"], O["ASSERT_VAL"], O["DJNZ"], O["DONE"], O["BIND"]]


@pytest.mark.skipif(not os.path.isdir("/root/reference/madsim"), reason="the reference checkout is not on this box")
def test_patch_context_lines_exist_in_the_reference():
    """bindings/rust/patches/0001-*.patch: every context / removed line of every hunk is a line of the reference file it patches."""
    patch = open(os.path.join(RS, "patches", "0001-builder-run_workload.patch")).read()
    cur, checked = None, 0
    for line in patch.splitlines():
        if line.startswith("+++ b/"):
            cur = open(os.path.join("/root/reference", line[6:])).read().splitlines()
            cur = {l.strip() for l in cur}
        elif cur is not None and (line.startswith(" ") or (line.startswith("-") and not line.startswith("---"))):
            if line[1:].strip():
                assert line[1:].strip() in cur, line
                checked += 1
        elif line.startswith("diff --git"):
            cur = None
    assert checked >= 15
