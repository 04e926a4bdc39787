"""tools/ref_twin: the reference-side pin kit.  No Rust here, so these tests keep the ORACLE side honest: every twin
workload is a valid table that passes, the oracle emits exactly the fields of schema.json, compare() accepts the
oracle's own record and rejects a perturbed one, and — the day tests/golden/ref_madsim.jsonl exists — real madsim's
fingerprints are compared for real."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_twin"))
import compare as CMP           # noqa: E402
import twin_workloads as T      # noqa: E402

import oracle                   # noqa: E402

SCHEMA = json.load(open(os.path.join(ROOT, "tools", "ref_twin", "schema.json")))
TYPES = {"integer": int, "number": (int, float), "string": str, "array": list, "null": type(None)}


def _check_schema(rec):
    assert set(SCHEMA["required"]) <= set(rec) <= set(SCHEMA["properties"])
    for k, v in rec.items():
        spec = SCHEMA["properties"][k]
        if "enum" in spec:
            assert v in spec["enum"], (k, v)
        else:
            ts = spec["type"] if isinstance(spec["type"], list) else [spec["type"]]
            assert isinstance(v, tuple(t for n in ts for t in (TYPES[n] if isinstance(TYPES[n], tuple) else (TYPES[n],)))), (k, v)


@pytest.mark.parametrize("name", sorted(T.ALL))
def test_twin_tables_pass_and_the_oracle_side_matches_the_schema(name):
    w = T.ALL[name]()
    out, summ = oracle.run_batch(w, 0, 32, T.config(name))
    if name in T.EXPECT_PANIC:                              # the reference test is #[should_panic]: verdict, no fingerprint tail
        assert (out["verdict"] == 1).all()
        rec = CMP.oracle_record(name, 5)
        _check_schema(rec)
        assert rec["verdict"] == "panic" and rec["elapsed_ns"] is None
        return
    assert summ.n_failed == 0, (name, out[out["verdict"] != 0][:1])
    assert len(set(out["obs_hash"].tolist())) == 32          # the trailing draw makes every seed's fingerprint distinct
    rec = CMP.oracle_record(name, 5, want_log=True)
    _check_schema(rec)
    assert rec["verdict"] == "pass" and rec["obs"][-2] == rec["elapsed_ns"] and rec["obs"][-1] < 2**32
    assert T.fold_obs(rec["obs"]) == int(out[5]["obs_hash"])
    assert 50 <= int(out[5]["clock_ns"]) - rec["elapsed_ns"] < 100      # the final poll's 50..100 ns (task/mod.rs:319-321)


def test_workload_names_match_schema_and_rust_source():
    names = set(SCHEMA["properties"]["workload"]["enum"])
    assert names == set(T.ALL)
    src = open(os.path.join(ROOT, "tools", "ref_twin", "src", "main.rs")).read()
    for n in names:
        assert f'"{n}"' in src, f"{n} missing from main.rs"


def test_known_fingerprints():
    rec = CMP.oracle_record("sleep_1s", 7)
    assert rec["elapsed_ns"] == 1_000_000_000 + 50        # deterministic_std_instant (time/system_time.rs:140-154)
    rec = CMP.oracle_record("pingpong4", 3)
    assert rec["msg_count"] == 2 * 2 * 64
    rec = CMP.oracle_record("pingpong4", 0, loss=0.05)
    _check_schema(rec)
    assert rec["verdict"] == "deadlock" and rec["elapsed_ns"] is None


def test_compare_accepts_the_oracle_record_and_rejects_perturbed_ones():
    ref = CMP.oracle_record("yield_order", 11, want_log=True)
    assert CMP.compare(json.loads(json.dumps(ref))) == []
    log = bytes.fromhex(ref["log_hex"])
    for k, v in (("msg_count", 1), ("verdict", "deadlock"), ("obs", ref["obs"][:-1] + [ref["obs"][-1] ^ 1]),
                 ("elapsed_ns", ref["elapsed_ns"] + 1), ("log_hex", (bytes([log[0] ^ 1]) + log[1:]).hex())):
        assert CMP.compare(dict(ref, **{k: v})), k


def test_real_madsim_fixture_if_present():
    path = os.path.join(ROOT, "tests", "golden", "ref_madsim.jsonl")
    if not os.path.exists(path):
        pytest.skip("no reference fixture yet: needs a Rust toolchain (tools/ref_twin/README.md)")
    bad = [(json.loads(l)["workload"], json.loads(l)["seed"], CMP.compare(json.loads(l))) for l in open(path) if l.strip()]
    assert not [b for b in bad if b[2]]
