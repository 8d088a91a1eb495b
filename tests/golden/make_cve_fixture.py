"""Golden vectors from the reference's own recorded run (data, not source): /root/reference/tests/output/
1761047573_v0_17/cve_search_results.csv holds, per query, the top-15 (item, score) lists of ONE index searched at
tau = 1.0 ("Cosine"), 0.8 ("Hybrid") and 0.62 ("Taumode") (tests/test_2_CVE_db.py:26-28, 250-270), scores printed to
6 decimals.  An item that appears in all three lists of a query gives a triple (s_1.0, s_0.8, s_0.62) of the SAME
(query, item) pair: with SPEC S11 (TAUMODE.md:33) score = tau*cos + (1-tau)*T the first two determine cos and T, and the
third is then a prediction to check.  Run here only (the reference does not travel): writes cve_blend.json."""
import csv
import json
import os
from collections import defaultdict

SRC = "/root/reference/tests/output/1761047573_v0_17/cve_search_results.csv"
HERE = os.path.dirname(os.path.abspath(__file__))
rows = defaultdict(dict)
for r in csv.DictReader(open(SRC)):
    rows[(int(r["query_id"]), r["cve_id"])][r["tau_method"]] = (int(r["rank"]), float(r["score"]))
out = defaultdict(list)
for (qid, item), m in sorted(rows.items()):
    if len(m) == 3:
        out[qid].append({"item": item, "s_1.0": m["Cosine"][1], "s_0.8": m["Hybrid"][1], "s_0.62": m["Taumode"][1],
                         "rank_0.62": m["Taumode"][0]})
doc = {"source": "tests/output/1761047573_v0_17/cve_search_results.csv of the reference (recorded run, 6-decimal scores)",
       "taus": {"Cosine": 1.0, "Hybrid": 0.8, "Taumode": 0.62}, "queries": {str(k): v for k, v in sorted(out.items())}}
json.dump(doc, open(os.path.join(HERE, "cve_blend.json"), "w"), indent=1)
print({k: len(v) for k, v in out.items()}, sum(len(v) for v in out.values()))
