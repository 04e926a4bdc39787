#!/usr/bin/env python3
"""Regenerate bindings/rust/madsim-hip-sys/src/lib.rs from include/madsim_hip.h (structs, constants, enum values, prototypes).
tests/test_rust_binding.py parses the committed lib.rs on its own and compares it with the header, so a hand edit that drifts
is caught whether or not this script is used.  Usage: python tools/gen_rust_sys.py > bindings/rust/madsim-hip-sys/src/lib.rs"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cheader as H  # noqa: E402

PRIM = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "int": "c_int",
        "double": "f64", "char": "c_char", "void": "c_void"}


def rust_type(ct):
    """C type string (tests/cheader.py form, e.g. 'const madsim_workload_t*', 'madsim_hip_ctx_t* const*') -> Rust."""
    ct = ct.strip()
    m = re.match(r"^(const\s+)?([\w ]+?)\s*((?:\*\s*(?:const\s*)?)*)$", ct)
    const0, base, stars = bool(m.group(1)), m.group(2).strip(), m.group(3)
    t = PRIM.get(base, base)
    ptrs = re.findall(r"\*\s*(const)?", stars)
    # innermost pointer's pointee constness is `const0`; each further level's pointee constness is the `const` after the star before it
    consts = [const0] + [bool(c) for c in ptrs[:-1]] if ptrs else []
    for c in consts:
        t = ("*const " if c else "*mut ") + t
    return t


def main():
    raw = open(H.HEADER_PATH).read()
    text = H.header_text()
    out = []
    w = out.append
    w("//! Raw FFI to `libmadsim_hip.so` — the MI355X (gfx950) many-seed runner behind madsim's `Builder::run`")
    w("//! (`madsim/src/sim/runtime/builder.rs:121-162`).  One item per item of `include/madsim_hip.h`, same names, same field")
    w("//! order; `tests/test_rust_binding.py` keeps the two in step without a Rust toolchain (field order, widths, constant")
    w("//! values, function arity and parameter types).  Regenerate with `python tools/gen_rust_sys.py`.")
    w("//!")
    w("//! Nothing here has a CPU fallback: without the library or a GPU every entry point returns an error code.")
    w("#![allow(non_camel_case_types, non_upper_case_globals)]")
    w("")
    w("use std::os::raw::{c_char, c_int, c_void};")
    w("")
    w("/// Opaque per-GPU runner state (`madsim_hip_ctx_t`).")
    w("#[repr(C)]")
    w("pub struct madsim_hip_ctx_t {")
    w("    _private: [u8; 0],")
    w("}")
    for name, fields in H.structs(text).items():
        w("")
        w("#[repr(C)]")
        w("#[derive(Clone, Copy, Debug)]")
        w(f"pub struct {name} {{")
        for fname, ctype, arr, ptr in fields:
            t = PRIM.get(ctype, ctype)
            if ptr:
                t = "*const " + t
            if arr:
                t = f"[{t}; {arr}]"
            rname = "r#match" if fname == "match" else fname
            w(f"    pub {rname}: {t},")
        w("}")
    w("")
    w("// ---- enum madsim_op / enum madsim_verdict -------------------------------------------------------------------------")
    for m in re.finditer(r"\b(MS_OP_\w+|MADSIM_(?:PASS|PANIC|DEADLOCK|TIME_LIMIT|OVERFLOW|STEP_LIMIT|UNSUPPORTED|INTERNAL))\s*=\s*(\d+)", text):
        ty = "u8" if m.group(1).startswith("MS_OP_") else "u32"
        w(f"pub const {m.group(1)}: {ty} = {m.group(2)};")
    w("")
    w("// ---- #define constants -------------------------------------------------------------------------------------------")
    for m in re.finditer(r"#define\s+(MADSIM_\w+)\s+\(?(0x[0-9a-fA-F]+|-?\d+)[uU]?\)?", raw):
        name, val = m.group(1), m.group(2)
        if name == "MADSIM_HIP_H":
            continue
        ty = "c_int" if name.startswith("MADSIM_E_") else "u32"
        w(f"pub const {name}: {ty} = {val};")
    w("")
    w('#[link(name = "madsim_hip")]')
    w('extern "C" {')
    for name, (ret, params) in H.functions(text, with_names=True).items():
        ps = ", ".join(f"{'r#' + n if n in ('match', 'type', 'ref') else n}: {rust_type(t)}" for t, n in params)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        w(f"    pub fn {name}({ps}){r};")
    w("}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
