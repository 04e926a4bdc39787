# Throw-away script behind DESIGN.md section 2 ("a wider soup"): random ops incl. address kinds, IP-less nodes, typed RPC, hooks, IPVS calls and panics on
# restarting nodes, host-compiled kernel vs oracle in both layouts.  Usage: PYTHONPATH=. python tools/experiment/op_soup_wide.py <programs> <base seed>
# Known open difference: program 1507471 of `... 8000 1500000` — a socket left bound by a restarted node's dead task, a receive registered on it by another task.
import random, sys, collections
import numpy as np
import oracle
from tests import emu, fuzz
from madsim_amd import _abi as A, workload as W

def gen(rng):
    wl = W.WorkloadBuilder()
    n = rng.randint(1, 3)
    nodes = [wl.create_node(restart_on_panic=rng.random() < 0.2, restart_on_panic_matching=(rng.choice(["1", "boom", 3]),) if rng.random() < 0.15 else (), ip=rng.random() > 0.1) for _ in range(n)]
    addrs = []
    for _ in range(rng.randint(1, 5)):
        addrs.append(wl.addr(nodes[rng.randrange(n)], rng.choice([1, 2]), ip=rng.choice(["node", "node", "unspecified", "loopback"])))
    named = [a for a in addrs if wl.socks[a].port != 0]
    if not named:
        named = [wl.addr(nodes[0], 1)]; addrs.append(named[0])
    vaddrs = [wl.virtual_addr(rng.randint(1, 2), 80) for _ in range(rng.randint(0, 2))]
    vaddrs = list(dict.fromkeys((wl.socks[v].node, v) for v in vaddrs).values()) if False else vaddrs
    services = []
    seen_v = set()
    for v in vaddrs:
        key = (wl.socks[v].node, wl.socks[v].port)
        if key in seen_v: continue
        seen_v.add(key)
        absent = rng.random() < 0.3
        services.append(wl.ipvs_service(v, [] if absent else [rng.choice(named) for _ in range(rng.randint(0, 3))], absent=absent))
    dsts = named + vaddrs
    ntask = rng.randint(1, 4)
    tasks = [wl.task(nodes[rng.randrange(n)], init=rng.random() < 0.1, pre=rng.random() < 0.1) for _ in range(ntask)]
    ops = ("try_bind,try_bind,bind,close,send,send,reply,recv_t,recv,sleep,sleep_until,mark,advance,yield,trace,tinst,loss,clog,unclog,spawn,spawn,join,"
           "connect,accept,csend,crecv,cclose,kill,restart,pause,resume,abort,rpc_call,rpc_call,rpc_recv,rpc_reply,hook_req,hook_rsp,"
           "ipvs_add_service,ipvs_del_service,ipvs_add_server,ipvs_del_server,panic,rand,randb,flag,assert_exit,spawn_mv").split(",")
    for ti, t in enumerate(tasks):
        t.mark(); t.set(0, rng.randint(1, 3)); top = t.label()
        for _ in range(rng.randint(2, 9)):
            op = rng.choice(ops); a = rng.choice(addrs); later = ti + 1 < len(tasks)
            if op == "bind": t.bind(a)
            elif op == "try_bind": t.try_bind(a); t.trace_val()
            elif op == "close": t.close(a)
            elif op == "send": t.send_to(a, rng.choice(dsts), rng.choice([1, 2]), rng.randrange(256))
            elif op == "reply": t.reply(a, rng.choice([1, 2]), rng.randrange(256))
            elif op == "recv": t.recv_from(a, rng.choice([1, 2])); t.trace_val()
            elif op == "recv_t": t.recv_from_timeout(a, rng.choice([1, 2]), ms=rng.choice([0, 1, 3, 10])); t.trace_val()
            elif op == "sleep": t.sleep(us=rng.choice([0, 10, 1000, 1500]))
            elif op == "sleep_until": t.sleep_until(ms=rng.choice([0, 1, 4]))
            elif op == "mark": t.mark()
            elif op == "advance": t.advance(us=rng.choice([0, 500, 2000]))
            elif op == "yield": t.yield_now()
            elif op == "trace": t.trace(rng.randrange(1000))
            elif op == "tinst": t.trace_instant()
            elif op == "loss": t.set_loss(rng.randrange(3))
            elif op == "clog": t.clog_node(rng.choice(nodes), rng.choice(["in", "out", "both"]))
            elif op == "unclog": t.unclog_node(rng.choice(nodes), rng.choice(["in", "out", "both"]))
            elif op == "connect": t.connect1(a, rng.choice(dsts)); t.trace_val()
            elif op == "accept": t.accept1(a)
            elif op == "csend": t.chan_send(rng.randrange(256)); t.trace_val()
            elif op == "crecv": t.chan_recv(); t.trace_val()
            elif op == "cclose": t.chan_close()
            elif op == "kill": t.kill(rng.choice(nodes))
            elif op == "restart": t.restart(rng.choice(nodes))
            elif op == "pause": t.pause(rng.choice(nodes))
            elif op == "resume": t.resume(rng.choice(nodes))
            elif op == "assert_exit": t.assert_exit(rng.choice(nodes), rng.random() < 0.5)
            elif op == "abort" and later: t.abort(tasks[rng.randrange(ti + 1, len(tasks))])
            elif op == "spawn" and later: t.spawn(tasks[rng.randrange(ti + 1, len(tasks))])
            elif op == "spawn_mv" and later: t.spawn(tasks[rng.randrange(ti + 1, len(tasks))], move_conn=rng.random() < 0.5, move_request=rng.random() < 0.5)
            elif op == "join" and later: t.join(tasks[rng.randrange(ti + 1, len(tasks))], expect_err=rng.random() < 0.2)
            elif op == "rpc_call": t.rpc_call(a, rng.choice(dsts), rng.randrange(2), rng.randrange(4), timeout_ms=rng.choice([0, 0, 5, 30])); t.trace_val()
            elif op == "rpc_recv": t.rpc_recv(a, rng.randrange(2)); t.trace_val()
            elif op == "rpc_reply": t.rpc_reply(a, rng.randrange(4))
            elif op == "hook_req": t.hook_rpc_req(rng.choice(nodes), rng.randrange(2), rng.choice([None, 1, 2]))
            elif op == "hook_rsp": t.hook_rpc_rsp(rng.choice(nodes), rng.choice([None, 1, 2]))
            elif op.startswith("ipvs") and services:
                sv = rng.choice(services)
                if op == "ipvs_add_service": t.ipvs_add_service(sv)
                elif op == "ipvs_del_service": t.ipvs_del_service(sv)
                elif op == "ipvs_add_server": t.ipvs_add_server(sv, rng.choice(named))
                else: t.ipvs_del_server(sv, rng.choice(named))
            elif op == "panic" and rng.random() < 0.3: t.panic(rng.choice([0, 1, 3, 13]))
            elif op == "rand": t.random_u32(); t.trace_val()
            elif op == "randb": t.rand_bool(rng.randrange(3)); t.trace_val()
            elif op == "flag": t.flag_add(rng.randrange(4), 1)
        t.djnz(0, top); t.done()
    m = wl.main()
    for t in tasks:
        if rng.random() < 0.8: m.spawn(t)
    if rng.random() < 0.5: m.sleep(ms=rng.randint(0, 5))
    for t in tasks:
        if rng.random() < 0.7: m.join(t, expect_err=rng.random() < 0.1)
    m.done()
    return wl.build(), A.Config.default(packet_loss_rate=rng.choice([0.0, 0.1]), loss_table=(0.0, 0.5, 1.0))

N = int(sys.argv[1]); base = int(sys.argv[2])
bad = 0; rej = 0; built = 0; verd = collections.Counter(); reasons = collections.Counter()
for k in range(N):
    rng = random.Random(base + k)
    try:
        w, cfg = gen(rng)
    except Exception as ex:
        rej += 1; reasons["build: " + str(ex)[:60]] += 1; continue
    built += 1
    for glob in (0, 1):
        print("RUN", k, glob, file=sys.stderr, flush=True)
        lim = fuzz.mixed_limits() if hasattr(fuzz, "mixed_limits") else fuzz.generous_limits()
        lim.max_tasks = 24
        if glob: lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL
        try:
            e = emu.run_batch(w, k * 3, 6, cfg, lim)
        except Exception as ex:
            rej += 1; reasons["emu: " + str(ex)[:70]] += 1; break
        o, _ = oracle.run_batch(w, k * 3, 6, cfg, lim)
        ok = (o == e) | (e["verdict"] == A.OVERFLOW)
        verd.update(o["verdict"].tolist())
        if not ok.all():
            bad += 1
            if bad <= 6: print("MISMATCH prog", base + k, "glob", glob, "\n  oracle", o[~ok][0], "\n  emu   ", e[~ok][0])
print("programs", N, "built", built, "rejected", rej, "mismatching", bad, "verdicts", dict(verd))
for r, c in reasons.most_common(8): print("  ", c, r)
