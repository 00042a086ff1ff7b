import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from test_gpu_consumers import _fitted
g, fun = _fitted(n=25, seed=3)
g.verbose = True
print("best so far", np.min(g.y), g.x[np.argmin(g.y[:, 0])])
np.random.seed(0)
print(g.BO(opt_type="min", opt_method="predict", method="EI", max_iter=4, predict_samps=3000, refine=True))
