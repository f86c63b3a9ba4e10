"""roofline.traffic / traffic_families are reproducible: the committed JSON files are exactly what tools/traffic_from_pmc.py
derives from the committed rocprofv3 --pmc summaries with its one stated unit rule, and the derivation itself is checked
against a hand computation from the CSV cells."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("traffic_from_pmc", os.path.join(ROOT, "tools", "traffic_from_pmc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_json_equals_recomputation():
    T = _tool()
    for kind, sp in T.SPEC.items():
        with open(os.path.join(ROOT, "profiles", sp["json"])) as f:
            committed = json.load(f)
        assert committed == json.loads(json.dumps(T.derive(kind))), kind


def test_unit_rule_by_hand():
    T = _tool()
    for kind, sp in T.SPEC.items():
        fetch, write = (T._means(os.path.join(ROOT, "profiles", n)) for n in sp["csv"])
        doc = T.derive(kind)
        for fam, (kernels, maps, slab_wgs) in sp["families"].items():
            by_hand = 0.0
            for k in kernels:
                f = [v for n, v in fetch.items() if n.startswith(k)]
                w = [v for n, v in write.items() if n.startswith(k)]
                assert len(f) == 1 and len(w) == 1, (k, f, w)
                by_hand += (2 * f[0] + w[0]) * 1024  # KB = 1024 B, reads doubled, every kernel of the family
            algo = maps * 32 * 128 * 128 * 64 * 4 + (2 * slab_wgs * 147456 + 147456 + 256 if slab_wgs else 0)
            got = doc["families_b32"][fam]
            assert abs(got["measured_MB"] * 1e6 - by_hand) <= 0.001 * by_hand, (fam, got, by_hand)
            assert abs(got["algorithmic_MB"] * 1e6 - algo) <= 0.001 * algo, (fam, got, algo)
            assert abs(got["ratio"] - by_hand / algo) < 1e-3
            assert 0.95 < got["ratio"] < 1.1, (fam, got)  # no wasted re-reads: every family moves its algorithmic bytes
