"""Same-box A/B of the symmetric-mixture memo (sampling.MixtureProposal): alternating runs of tools/bench_mh_chain.py with the memo on / off."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = ("import sys, os; sys.path.insert(0, %r); import torch; import gingr_amd.sampling as sp\n"
        "if os.environ.get('MEMO') == '0':\n"
        "    for c in (sp.RandomShapeUpdateProposal, sp.GaussianAxisRotationProposal, sp.GaussianAxisTranslationProposal): c.symmetric = False\n"
        "sys.argv = ['bench_mh_chain.py', '300', '0']\n"
        "exec(compile(open(%r).read(), 'bench_mh_chain.py', 'exec'), {'__file__': %r, '__name__': '__main__'})\n") % (
            ROOT, os.path.join(ROOT, "tools", "bench_mh_chain.py"), os.path.join(ROOT, "tools", "bench_mh_chain.py"))
res = {"0": [], "1": []}
for rep in range(4):
    for memo in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MEMO=memo), capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        res[memo].append(round(json.loads(line)["steps_per_s"]))
print("memo off", res["0"], "memo on", res["1"])
