"""A small parser for include/madsim_hip.h: struct fields (name, C type, array length) and function prototypes
(name, return type, parameter types).  Shared by the ABI tests that keep the ctypes mirror (madsim_amd/_abi.py) and the
Rust binding (bindings/rust/madsim-hip-sys/src/lib.rs) in step with the header — no compiler needed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER_PATH = os.path.join(ROOT, "include", "madsim_hip.h")

_SIZES = {"uint8_t": 1, "uint16_t": 2, "uint32_t": 4, "uint64_t": 8, "int64_t": 8, "int": 4, "double": 8, "float": 4, "char": 1}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def header_text():
    return _strip_comments(open(HEADER_PATH).read())


def structs(text=None):
    """{typedef name: [(field, ctype, array_len or 0, is_pointer)]} in declaration order."""
    text = text or header_text()
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        body, name = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            dm = re.match(r"(const\s+)?([\w ]+?)\s*(\*?)\s*(\w+(?:\s*,\s*\w+)*)\s*(?:\[(\d+)\])?$", decl)
            assert dm, f"cannot parse field {decl!r} of {name}"
            ctype, ptr, names, arr = dm.group(2).strip(), bool(dm.group(3)), dm.group(4), int(dm.group(5) or 0)
            for fname in names.split(","):
                fields.append((fname.strip(), ctype, arr, ptr))
        out[name] = fields
    return out


def layout(fields, struct_sizes=None):
    """(offsets, size) of a struct under the C ABI of x86-64 / aarch64 (natural alignment)."""
    off, align_max, offs = 0, 1, {}
    for fname, ctype, arr, ptr in fields:
        size = 8 if ptr else _SIZES.get(ctype) or (struct_sizes or {})[ctype][0]
        align = 8 if ptr else _SIZES.get(ctype) or (struct_sizes or {})[ctype][1]
        off = (off + align - 1) // align * align
        offs[fname] = off
        off += size * (arr or 1)
        align_max = max(align_max, align)
    return offs, (off + align_max - 1) // align_max * align_max


def functions(text=None, with_names=False):
    """{name: (return type, [parameter type strings])} for every prototype in the header (with_names: [(type, name)])."""
    text = text or header_text()
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"enum\s+\w+\s*\{.*?\}\s*;", " ", text, flags=re.S)
    text = re.sub(r"^[ \t]*#.*$", " ", text, flags=re.M)            # preprocessor lines (a #define just before a prototype is not its return type)
    out = {}
    for m in re.finditer(r"([\w\*\s]+?)\b(madsim_\w+)\s*\(([^()]*)\)\s*;", text):
        ret = " ".join(m.group(1).replace("extern", "").split())
        params = []
        for p in m.group(3).split(","):
            p = " ".join(p.split())
            if p in ("", "void"):
                continue
            pm = re.match(r"(.*?)(\w+)$", p)                    # drop the parameter name
            t = pm.group(1).strip() if pm and not p.endswith("*") else p
            t = " ".join(t.replace(" *", "*").split())
            params.append((t, pm.group(2) if pm and not p.endswith("*") else f"a{len(params)}") if with_names else t)
        out[m.group(2)] = (ret.replace(" *", "*"), params)
    return out


def defines(text=None):
    text = text or open(HEADER_PATH).read()
    return {m.group(1): m.group(2) for m in re.finditer(r"#define\s+(MADSIM_\w+|MS_\w+)\s+\(?(-?\w+)\)?", text)}
