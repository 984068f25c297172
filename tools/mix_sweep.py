"""Order of the work queue: chunks of the thick atmospheres alternating with a share of the thin ones (experiment
build, tools/ablation: RT_AMD_MIX=tau_split,beta) against the product's order, on one rank and on the share of one
of 8.  python tools/mix_sweep.py <abl.so> [tau_split,beta ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.abspath(sys.argv[1])
rules = sys.argv[2:] or ["0.1,1.0", "0.1,0.7", "0.1,0.5", "0.3,0.7", "0.3,1.0", "0.03,0.7", "1.0,0.7"]
for _ in range(2):
    for rule in ["product"] + rules:
        e = dict(os.environ, RT_AMD_LIB=lib, RT_AMD_ALLOW_ABLATION="1")
        if rule != "product":
            e["RT_AMD_MIX"] = rule
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_ranks.py"), "20000000", os.environ.get("MIX_WORLDS", "8")], env=e, capture_output=True, text=True)
        if out.returncode:
            print(rule, "failed", out.stderr[-400:])
            sys.exit(1)
        w = {l.split(":")[0]: l for l in out.stdout.splitlines() if l.startswith("W=")}
        print(f"{rule:10s} {w['W=1']}   " + "   ".join(k + " max" + v.split("max")[1].split("projected")[0] for k, v in w.items() if k != "W=1"), flush=True)
