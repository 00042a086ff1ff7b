"""MAP driver: the host half of pm.find_MAP as GPMCMC.__fit uses it (gpmcmc.py:326-346,357).
[3P] pymc/tuning/starting.py: scipy.optimize.minimize(method="L-BFGS-B", jac=True) on the negative
joint log-posterior in the transformed (unconstrained) space, started from the model's initial
point, at most ``maxeval`` (5000) evaluations; the result is the best point seen."""
import numpy as np
from scipy.optimize import minimize


class _StopOnMaxEval(Exception):
    pass


def find_MAP(logp_dlogp, q0, method="L-BFGS-B", maxeval=5000, progressbar=False, **kwargs):
    """Maximise ``logp_dlogp(q) -> (logp, grad)``.  Returns (q_best, info dict)."""
    if method != "L-BFGS-B":
        raise ValueError("only L-BFGS-B (PyMC's default for differentiable models) is provided")
    state = {"n": 0, "best": (-np.inf, np.array(q0, dtype=np.float64))}

    def cost(q):
        if state["n"] >= maxeval:
            raise _StopOnMaxEval
        state["n"] += 1
        v, g = logp_dlogp(q)
        if np.isfinite(v) and v > state["best"][0]:
            state["best"] = (v, q.copy())
        if not np.isfinite(v):
            return 1.0e100, np.zeros_like(q)  # PyMC: non-finite cost -> large value, optimiser backs off
        if progressbar:
            print(f"  eval {state['n']:4d}: logp = {v:,.5g}, ||grad|| = {np.linalg.norm(g):,.5g}")
        return -v, -g

    try:
        res = minimize(cost, np.array(q0, dtype=np.float64), method="L-BFGS-B", jac=True, **kwargs)
        msg = res.message
    except _StopOnMaxEval:
        msg = "maxeval reached"
    best_v, best_q = state["best"]
    return best_q, {"logp": best_v, "nfev": state["n"], "message": msg}
