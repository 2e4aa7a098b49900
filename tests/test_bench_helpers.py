"""Host-side pieces of bench.py that need no GPU: the profiler counters are tied to the library build they were collected
with, and the byte model of the reference-order traversal adds up as documented."""
import importlib.util
import json
import os
import types

from cuda_raytracing_amd import abi
from helpers import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_counters_of_another_build_are_refused(tmp_path, monkeypatch):
    bench = _bench()
    cur = tmp_path / "profiles" / "current"
    cur.mkdir(parents=True)
    (cur / "pmc_headline.json").write_text(json.dumps({"build_id": "aaaaaaaaaaaaaaaa", "pmc_sq": {"SQ_INSTS_VALU": 1.0}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    pmc, path, stale = bench.profile_counters("headline", "aaaaaaaaaaaaaaaa")
    assert pmc is not None and not stale and path.endswith("pmc_headline.json")
    pmc, path, stale = bench.profile_counters("headline", "bbbbbbbbbbbbbbbb")
    assert pmc is None and stale and path.endswith("pmc_headline.json")  # bench.py then prints frac: null, counters_stale: true
    pmc, path, stale = bench.profile_counters("c3", "aaaaaaaaaaaaaaaa")     # no file at all: nothing to use, nothing stale
    assert pmc is None and path is None and not stale
    (cur / "pmc_c3.json").write_text(json.dumps({"pmc_sq": {"SQ_INSTS_VALU": 1.0}}))  # a file from before build ids existed
    pmc, _, stale = bench.profile_counters("c3", "aaaaaaaaaaaaaaaa")
    assert pmc is None and stale


def test_library_reports_a_build_id():
    lib = abi.load_library()
    bid = lib.rt_build_id().decode()
    assert len(bid) == 16 and all(c in "0123456789abcdef" for c in bid)


def test_committed_counters_name_one_build():
    """profiles/current/ holds the counter files bench.py reads: all from the same build of the library."""
    cur = os.path.join(ROOT, "profiles", "current")
    ids = set()
    for name in os.listdir(cur):
        if name.startswith("pmc_") and name.endswith(".json"):
            with open(os.path.join(cur, name)) as f:
                ids.add(json.load(f).get("build_id"))
    assert len(ids) == 1 and None not in ids, ids


def test_byte_model_of_the_reference_order_traversal():
    bench = _bench()
    s = types.SimpleNamespace(nodes_visited=10, tris_tested=3, rays=5, shaded_hits=2, texture_fetches=7, pixels=11)
    assert bench.algorithmic_bytes(s) == 64 * 10 + 72 * 3 + 48 * 5 + 128 * 2 + 16 * 7 + 4 * 11
    c = {"rays_traced": 5, "nodes_visited": 10, "tris_tested": 3, "shaded_hits": 2, "texture_fetches": 7, "n_prims": 4, "n_lights": 2}
    assert bench.flops_model(c) == 120 * 10 + 60 * 3 + 40 * 5 * 4 + 50 * 2 * 2 + 25 * 7
