"""Regression pins of every CPU oracle (tests/golden/oracle_digests.json, made by scripts/make_golden.py): the oracles
are what all GPU parity tests compare against, so a silent change to one must fail here, on the CPU."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_oracle_reproduces_its_committed_digest(oracle, each_canon):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "scripts", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")) as f:
        d = json.load(f)
    # top level: canon 0 (one rounding per operator; unchanged since round 5), "_canon1": the contracted form
    want = d["_canon1"] if each_canon else {k: v for k, v in d.items() if not k.startswith("_")}
    got = mg.digests()
    assert set(got) == set(want)
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, bad
