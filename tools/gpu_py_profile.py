"""cProfile of the host side of a headline run (diagnostic)."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
opt = PCA_BO(budget=450, n_DoE=120, random_seed=15400, maximization=False)
prob = BBOBProblem(15, 0, 40)
opt._start(prob)
for _ in range(20): opt._bo_iteration(prob)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): opt._bo_iteration(prob)
pr.disable(); opt._finish()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
