#!/usr/bin/env python3
"""Static instruction budget of one sim_kernel build by SOURCE LINE (no GPU needed).

Compiles sim_kernel.hip for ONE variant with line tables, reads the `.loc` directives of the assembly and counts, per source line,
the VALU / SALU / LDS / VMEM / branch instructions the compiler made of it.  A wave executes every line some lane of it reaches, so for
the straight-line part of a pass (everything but the rejection loops) the static count IS the per-pass cost; loops count once here.

    python tools/isa_lines.py ['X(false,false,6,MADSIM_FEAT_COMPACT,true,false)'] [top=45]
"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var = sys.argv[1] if len(sys.argv) > 1 else "X(false,false,6,MADSIM_FEAT_COMPACT,true,false)"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
out = "/tmp/isa_lines.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Os", "-std=c++17", "-fPIC", "-Wno-unused-function", "-gline-tables-only",
                       f"-DMADSIM_FOR_EACH_VARIANT(X)={var}", "--cuda-device-only", "-S", "sim_kernel.hip", "-o", out],
                      cwd=os.path.join(ROOT, "madsim_amd", "csrc"), stderr=subprocess.DEVNULL)
files, cur, in_kernel = {}, None, False
cnt = collections.defaultdict(lambda: collections.Counter())
tot = collections.Counter()
site_cnt = collections.defaultdict(lambda: collections.Counter()); cur_site = None
for ln in open(out):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)
    if m:
        files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)); continue
    if s.startswith("_ZN8madsim_k10sim_kernel") and ":" in s.split(";")[0]:
        in_kernel = True; continue
    if s.startswith(".Lfunc_end"):
        in_kernel = False
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        # the inline chain in the comment: innermost first; `site` = the frame inside poll_task / sim_kernel that the code belongs to
        chain = re.findall(r"([\w./-]+):(\d+):\d+", s.split(";", 1)[1]) if ";" in s else []
        site = None
        for fn, l in chain:
            b = os.path.basename(fn)
            if b in ("k_poll.h", "k_main.h"):
                site = (b, int(l)); break
        cur_site = site or (os.path.basename(chain[0][0]), int(chain[0][1])) if chain else None
        continue
    if not in_kernel or not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    k = "branch" if op.startswith(("s_cbranch", "s_branch")) else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else \
        "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu" if op.startswith("s_") else "other"
    cnt[cur][k] += 1; tot[k] += 1
    if cur_site: site_cnt[cur_site][k] += 1
print(f"variant {var}: {dict(tot)}")
src = {}
def text(f, l):
    p = files.get(f, "?")
    p2 = p if os.path.isabs(p) else os.path.normpath(os.path.join(ROOT, "madsim_amd", "csrc", p))
    if p2 not in src:
        try: src[p2] = open(p2).read().split("\n")
        except OSError: src[p2] = []
    t = src[p2][l - 1].strip() if 0 < l <= len(src[p2]) else ""
    return os.path.basename(p) + f":{l}", t[:110]
byfile = collections.defaultdict(collections.Counter)
for (f, l), c in cnt.items():
    for k, v in c.items(): byfile[os.path.basename(files.get(f, '?'))][k] += v
print("by file:")
for f, c in sorted(byfile.items(), key=lambda kv: -kv[1]["valu"]):
    print(f"  {f:22s} valu {c['valu']:5d} salu {c['salu']:5d} branch {c['branch']:4d} lds {c['lds']:4d} vmem {c['vmem']:3d} wait {c['wait']:4d}")
print(f"top {top} source lines by VALU + SALU:")
for (f, l), c in sorted(cnt.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"]))[:top]:
    w, t = text(f, l)
    print(f"  {w:18s} valu {c['valu']:4d} salu {c['salu']:4d} br {c['branch']:3d} lds {c['lds']:3d} | {t}")

print(f"top {top} sites (the line of k_poll.h / k_main.h whose inlined callees the instructions belong to):")
def stext(b, l):
    p2 = os.path.join(ROOT, "madsim_amd", "csrc", "kernel", b)
    if p2 not in src:
        try: src[p2] = open(p2).read().split("\n")
        except OSError: src[p2] = []
    return src[p2][l - 1].strip()[:120] if 0 < l <= len(src[p2]) else ""
for (b, l), c in sorted(site_cnt.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"]))[:top]:
    print(f"  {b + ':' + str(l):16s} valu {c['valu']:4d} salu {c['salu']:4d} br {c['branch']:3d} lds {c['lds']:3d} | {stext(b, l)}")
