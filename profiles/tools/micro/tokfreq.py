import os, sys, numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
import bench
bench.GEN_WORKERS = 0
data = bench.generate([("c3", "im150_t0", 100_000_000, 20240002)])["c3"]
from imcoalhmm_amd.hmm import Forwarder
fw = Forwarder.from_array(data, 3)
tok = fw.new_obs
print("tokens", len(tok), "alphabet", fw.new_nsyms, "columns/token", len(data) / len(tok))
cnt = np.bincount(tok, minlength=fw.new_nsyms).astype(np.float64)
order = np.argsort(-cnt)
cum = np.cumsum(cnt[order]) / cnt.sum()
for k in (1, 2, 5, 10, 20, 40, 80, 160, 320, 640, 1280, 2560):
    print("top %4d tokens cover %.3f of the steps (%.1f MB of operators at 180 KB)" % (k, cum[k - 1], k * 0.18))
