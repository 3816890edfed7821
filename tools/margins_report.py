"""Summarise VPTR_MARGIN_LOG files of `pytest -m gpu` runs: explicit (value, bound) checks of the self-comparison tests, and per test
the largest helpers.rel() value (the parity bar of those tests is 1e-3 unless the test says otherwise).

    python tools/margins_report.py gpurun_out/r04/margins_*.log
"""
import json
import sys

checks, rels = {}, {}
for path in sys.argv[1:]:
    for line in open(path):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if "check" in r:
            c = checks.setdefault(r["check"], [0.0, r["bound"], 0])
            c[0] = max(c[0], r["value"]); c[2] += 1
        elif "rel" in r:
            t = rels.setdefault(r["test"], [0.0, 0])
            t[0] = max(t[0], r["rel"]); t[1] += 1
print("## explicit checks (worst value over %d log files)" % len(sys.argv[1:]))
print("| check | worst value | bound | bound / worst | samples |\n|---|---|---|---|---|")
for k, (v, b, n) in sorted(checks.items(), key=lambda kv: -(kv[1][0] / kv[1][1])):
    print("| %s | %.3g | %.3g | %.0fx | %d |" % (k, v, b, b / max(v, 1e-300), n))
print("\n## largest rel-L2 per test (helpers.rel; tests with a worst value above 1e-4 listed)")
print("| test | worst rel-L2 | rel() calls |\n|---|---|---|")
for k, (v, n) in sorted(rels.items(), key=lambda kv: -kv[1][0]):
    if v > 1e-4:
        print("| %s | %.3g | %d |" % (k, v, n))
print("\n%d tests logged rel() values; %d of them stay below 1e-4" % (len(rels), sum(1 for v, _ in rels.values() if v <= 1e-4)))
