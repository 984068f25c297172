"""Order of the work queue for short launches (the share of one of 8 ranks): interleaving threshold and length of
the ordered tail, experiment build (tools/ablation).  python tools/tail_sweep.py <abl.so>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.abspath(sys.argv[1])
cases = [("product", {}), ("ilv4 tail 2/8", {"RT_AMD_ILV": "4", "RT_AMD_TAIL": "2,8"}), ("ilv4 tail 2/3", {"RT_AMD_ILV": "4", "RT_AMD_TAIL": "2,3"}),
         ("ilv4 tail 3/2", {"RT_AMD_ILV": "4", "RT_AMD_TAIL": "3,2"}), ("ilv4 tail 4/2", {"RT_AMD_ILV": "4", "RT_AMD_TAIL": "4,2"}),
         ("ilv4 tail 1/8", {"RT_AMD_ILV": "4", "RT_AMD_TAIL": "1,8"})]
for _ in range(2):
    for name, env in cases:
        e = dict(os.environ, RT_AMD_LIB=lib, RT_AMD_ALLOW_ABLATION="1", **env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_ranks.py"), "20000000", "8"], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("W=8")]
        w1 = [l for l in out.stdout.splitlines() if l.startswith("W=1")]
        print(f"{name:16s} {w1[0] if w1 else ''}  {line[0] if line else out.stderr[-300:]}", flush=True)
