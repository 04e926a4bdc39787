#!/usr/bin/env python3
"""Memory-site model of the global-state builds of sim_kernel, on CPU (no GPU needed).

Builds the host emulation of the device code (tests/emu) unoptimised with frame pointers, keys every global-memory access by
the chain of return addresses above it (on the GPU everything is inlined: one chain = one machine instruction site) and
counts, per site, how often a 64-lane wave issues the instruction per main-loop iteration (max over lanes: lanes re-converge
at the loop top) and how many lanes take part.  With tools/ubench_vmem.hip's cost of a wave-instruction by active lanes this
prices the kernel's memory pipeline time site by site.

    python tools/mem_site_model.py raft [seeds=128] [top=40]
"""
import ctypes as C, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from madsim_amd import workload, _abi as A

EMU = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU, "libmadsim_emu_sites.so")
subprocess.check_call(["g++", "-O0", "-g", "-fno-omit-frame-pointer", "-fno-inline", "-std=c++17", "-fPIC", "-shared", "-DMADSIM_EMU",
                       "-DMADSIM_EMU_REGIONS", "-DMADSIM_EMU_SITES", "-x", "c++", "-I" + EMU, "-o", LIB, os.path.join(EMU, "emu_driver.cpp")])
L = C.CDLL(LIB)
L.madsim_emu_run_batch.argtypes = [C.POINTER(A.Workload), C.POINTER(A.Config), C.c_uint64, C.c_uint64, C.POINTER(A.Limits),
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
which = sys.argv[1] if len(sys.argv) > 1 else "raft"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 128
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
w, lim = {"kv": (workload.kv_rpc, workload.kv_rpc_limits), "raft": (workload.raft_election, workload.raft_election_limits),
          "topo": (workload.streaming_topology, workload.streaming_topology_limits)}[which]
w, lim = w(), lim()
lim.state_mem |= int(os.environ.get("MADSIM_MODEL_STATE_FLAGS", "0"), 0)      # e.g. 0x100 = MADSIM_STATE_DEDUP_TIMERS
cfg = A.Config.default()
out = np.zeros(count, dtype=A.RESULT_DTYPE)
rc = L.madsim_emu_run_batch(w.ref(), C.byref(cfg), 0, count, C.byref(lim), out.ctypes.data_as(C.c_void_p), 1, None, 0, None)
assert rc == 0, rc
trips = (C.c_double * 32)(); visits = (C.c_double * 32)(); iters = C.c_double()
L.madsim_emu_region_stats(trips, visits, C.byref(iters))
it = iters.value
steps = int(out["steps"].sum())

# map return addresses to source lines: the library's load base from /proc/self/maps
base = None
for line in open("/proc/self/maps"):
    if "libmadsim_emu_sites.so" in line:
        lo = int(line.split("-")[0], 16)
        base = lo if base is None else min(base, lo)
n = L.madsim_emu_site_count()
L.madsim_emu_site_zero.restype = C.c_uint64; L.madsim_emu_site_zero.argtypes = [C.c_uint32]
L.madsim_emu_site.argtypes = [C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
sites = []
addrs = set()
for i in range(n):
    ra = (C.c_size_t * 7)(); kind = C.c_int(); t = C.c_double(); v = C.c_double()
    L.madsim_emu_site(i, ra, C.byref(kind), C.byref(t), C.byref(v))
    ras = [int(x) - base for x in ra if x]
    sites.append((ras, kind.value, t.value, v.value, int(L.madsim_emu_site_zero(i)))); addrs.update(ras)
addrs = sorted(addrs)
res = subprocess.run(["addr2line", "-f", "-C", "-e", LIB] + [hex(a - 1) for a in addrs], capture_output=True, text=True).stdout.splitlines()
sym = {}
for k, a in enumerate(addrs):
    fn, loc = res[2 * k], res[2 * k + 1]
    fn = fn.replace("madsim_k::", "").split("(")[0].split("<")[0]
    sym[a] = f"{fn}:{os.path.basename(loc).split(' ')[0]}"


def cost(lanes):           # CU cycles per wave-instruction by active lanes on own lines (profiles/r3_vmem_cost.txt, L2-resident .. HBM)
    pts = [(1, 17, 50), (4, 18, 50), (16, 90, 190), (64, 165, 390)]
    for (l0, a0, b0), (l1, a1, b1) in zip(pts, pts[1:]):
        if lanes <= l1:
            f = (lanes - l0) / (l1 - l0) if l1 > l0 else 0
            return a0 + f * (a1 - a0), b0 + f * (b1 - b0)
    return pts[-1][1:]


HELPERS = ("URef", "WRef", "gs_load32", "gs_store32", "gs_load128", "gs_store128", "spill_load", "spill_store", "operator",
           "buf_load32", "buf_store32", "buf_load128", "buf_store128", "insn_fetch", "make_uref", "make_wref")
KIND = ["ld32", "st32", "ld128", "st128"]
tot_t = sum(s[2] for s in sites); tot_v = sum(s[3] for s in sites)
print(f"# tools/mem_site_model.py {which} {count}: {count} seeds, {it:.0f} wave-iterations, {steps} executor steps "
      f"({steps / it:.1f} lane-steps per wave-iteration)")
print(f"# {n} machine sites; wave memory instructions per wave-iteration {tot_t / it:.1f}, lane accesses per wave-iteration {tot_v / it:.1f} "
      f"(avg {tot_v / tot_t:.1f} lanes per instruction); per lane-step: {tot_t / steps:.2f} wave-instructions, {tot_v / steps:.2f} lane accesses")
c_lo = sum(s[2] * cost(s[3] / s[2])[0] for s in sites if s[2]); c_hi = sum(s[2] * cost(s[3] / s[2])[1] for s in sites if s[2])
print(f"# priced with ubench_vmem: {c_lo / it:.0f} .. {c_hi / it:.0f} CU cycles of memory pipeline per wave-iteration "
      f"({c_lo / steps:.0f} .. {c_hi / steps:.0f} per lane-step); fully converged (64 lanes per instruction): "
      f"{tot_v / 64 * 165 / steps:.0f} .. {tot_v / 64 * 390 / steps:.0f} per lane-step")
print(f"{'trips/iter':>10s} {'lanes':>6s} {'cyc/iter(L2)':>12s}  kind   zero  site (innermost first; zero = share of the loads that read 0: a mirror bit could skip them)")
for ras, kind, t, v, z in sorted(sites, key=lambda s: -s[2] * cost(s[3] / max(s[2], 1))[0])[:top]:
    chain = " < ".join(sym[a] for a in ras[:6] if sym[a].split(":")[0].strip() not in HELPERS or True)
    zs = f"{z / v:5.2f}" if kind in (0, 2) and v else "    -"
    print(f"{t / it:10.3f} {v / t:6.1f} {t * cost(v / t)[0] / it:12.0f}  {KIND[kind]:6s} {zs} {chain}")
by_fn = collections.Counter(); by_fn_t = collections.Counter(); by_fn_v = collections.Counter()
for ras, kind, t, v, _z in sites:
    if not t: continue
    names = [sym[a].split(":")[0].strip() for a in ras]
    key = next((x for x in names if not any(x.startswith(h) for h in HELPERS)), names[0])
    by_fn[key] += t * cost(v / t)[0]; by_fn_t[key] += t; by_fn_v[key] += v
print("# by enclosing function: cycles per wave-iteration (L2 pricing), wave-instructions per wave-iteration, lanes per instruction")
for k, c in by_fn.most_common(30):
    print(f"{c / it:10.0f} {by_fn_t[k] / it:8.2f} {by_fn_v[k] / by_fn_t[k]:6.1f}  {k}")
