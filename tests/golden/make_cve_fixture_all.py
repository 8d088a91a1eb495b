"""Golden vectors from ALL of the reference's recorded CVE runs (data, not source): the six directories under
/root/reference/tests/output/ that hold a cve_search_results.csv (crate versions 0.15, 0.16, 0.17; three queries each,
one index per run, searched at tau = 1.0 "Cosine", 0.8 "Hybrid", 0.62 "Taumode": tests/test_2_CVE_db.py:26-28,250-270;
scores printed to 6 decimals) and the lambda_q each query's search printed into test_results.txt
("[pyarrowspace] search: qlen=..., lambda_q=...", src/lib.rs:155).  Per run and query: the ranked (item, score) lists at
the three taus, lambda_q, and the triples (s_1.0, s_0.8, s_0.62) of the items that appear in all three lists.
What the numbers pin is the FORM of the scorer (SPEC S11 = TAUMODE.md:33); the embeddings behind them are not in the tree.
Run here only (the reference does not travel): writes cve_blend_all.json."""
import csv
import json
import os
import re
from collections import defaultdict

ROOT = "/root/reference/tests/output"
HERE = os.path.dirname(os.path.abspath(__file__))
TAUS = {"Cosine": 1.0, "Hybrid": 0.8, "Taumode": 0.62}
runs = {}
for d in sorted(os.listdir(ROOT)):
    src = os.path.join(ROOT, d, "cve_search_results.csv")
    if not os.path.isfile(src):
        continue
    lists = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(src)):
        lists[int(r["query_id"])][r["tau_method"]].append((int(r["rank"]), r["cve_id"], float(r["score"])))
    # lambda_q per query, in the order the queries ran (three identical lines per query: one per tau)
    lq = []
    txt = open(os.path.join(ROOT, d, "test_results.txt")).read()
    for block in re.split(r"={20,}\nQuery \d+:", txt)[1:]:
        m = re.findall(r"lambda_q=([0-9.eE+-]+)", block)
        lq.append(float(m[0]) if m else None)
    queries = {}
    for qi, (qid, by) in enumerate(sorted(lists.items())):
        ranked = {k: [(c, s) for _, c, s in sorted(v)] for k, v in by.items()}
        common = set(c for c, _ in ranked["Cosine"]) & set(c for c, _ in ranked["Hybrid"]) & set(c for c, _ in ranked["Taumode"])
        sc = {k: dict(v) for k, v in ranked.items()}
        triples = [{"item": c, "s_1.0": sc["Cosine"][c], "s_0.8": sc["Hybrid"][c], "s_0.62": sc["Taumode"][c]}
                   for c, _ in ranked["Taumode"] if c in common]
        queries[str(qid)] = {"lambda_q": lq[qi] if qi < len(lq) else None, "lists": ranked, "triples": triples}
    runs[d] = queries
doc = {"source": "tests/output/*/cve_search_results.csv + test_results.txt of the reference (recorded runs, 6-decimal scores)",
       "taus": TAUS, "runs": runs}
json.dump(doc, open(os.path.join(HERE, "cve_blend_all.json"), "w"), indent=1)
print({d: {q: (v["lambda_q"], len(v["triples"])) for q, v in qs.items()} for d, qs in runs.items()})
