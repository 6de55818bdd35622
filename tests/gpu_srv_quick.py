import sys, os, time
sys.path.insert(0, 'para-ortho-pca-bo_amd')
import numpy as np, torch
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
torch.set_num_threads(4)
import hashlib
opt = PCA_BO(budget=60, n_DoE=30, random_seed=15101, maximization=False)
t=time.time(); opt(BBOBProblem(15, 1, 10)); dt=time.time()-t
print(os.environ.get("PCABO_ACQ_SERVER","default"), "d10 ok", dt, hashlib.md5(np.array(opt.f_evals).tobytes()).hexdigest(), flush=True)
