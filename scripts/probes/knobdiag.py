import os, subprocess, sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests.test_gpu_giant import CHILD, ROOT, LENGTHS
def run(tag, env):
    out = "/tmp/kd_%s.npy" % tag
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out, k=100, prec=False, repeat=1, maxupd=1500, w=1.0, niter=2)], check=True, env=e, cwd=ROOT, capture_output=True, text=True)
    print(tag, [l for l in r.stdout.splitlines() if l.startswith("PLAN")][0][:300]); print(r.stderr[-300:])
    return np.load(out)[0]
import hashlib
res = {}
for rep in range(int(os.environ.get("KD_REPS", "3"))):
    for tag, env in (("gave", {"POISMF_HIP_TEAM_SPIN_LIMIT": "1"}), ("one", {"POISMF_HIP_NO_LANE_TEAMS": "1", "POISMF_HIP_NO_GIANT_TEAMS": "1"}), ("team", {})):
        out = "/tmp/kd_%s.npy" % tag
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out, k=100, prec=False, repeat=1, maxupd=1500, w=1.0, niter=2)], check=True, env=e, cwd=ROOT, capture_output=True, text=True)
        r = np.load(out)[0]
        res.setdefault(tag, []).append(hashlib.sha256(r.tobytes()).hexdigest()[:10])
for tag, hs in res.items():
    print(tag, hs)
