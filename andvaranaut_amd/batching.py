"""Rendezvous of several host threads on ONE device handle: the NUTS chains that share a GPU (pm.sample's chains,
gpmcmc.py:351) each ask for LML + gradient at their own theta once per leapfrog step; this layer collects one request
from every active chain and issues ONE batched device call (MiGP.lml_grad_batch -> mi_gp_lml_grad_batch: every kernel
launch carries blockIdx.z = problem), instead of one handle, two streams and one host thread per chain.  A batched
evaluation returns the same bits as the one-at-a-time entry point, so every chain's draws are those of the unbatched
schedule whatever the other chains do."""
import threading

import numpy as np


class BatchedEvaluator:
    def __init__(self, batch_fn, nclients):
        """batch_fn(thetas (k, ntheta)) -> (values (k,), grads (k, ntheta)); nclients threads will call register() once,
        evaluate() any number of times, then leave()."""
        self._fn = batch_fn
        self._cond = threading.Condition()
        self._active = int(nclients)
        self._pending = []      # thetas of the current round, in arrival order
        self._round = 0
        self._results = {}      # round -> (values, grads)
        self._taken = {}        # round -> results still to be picked up
        self._error = None
        self.rounds = 0
        self.evaluations = 0

    def _run(self):
        """caller holds the condition; runs the pending round and wakes the waiters"""
        thetas = np.array(self._pending)
        try:
            vals, grads = self._fn(thetas)
            self._results[self._round] = (np.asarray(vals), np.asarray(grads))
        except Exception as e:  # noqa: BLE001 - re-raised in every waiting client
            self._error = e
            self._results[self._round] = None
        self._taken[self._round] = len(self._pending)
        self.rounds += 1
        self.evaluations += len(self._pending)
        self._pending = []
        self._round += 1
        self._cond.notify_all()

    def evaluate(self, theta):
        """(value, gradient) at theta; blocks until every active client has asked (or left)."""
        with self._cond:
            rnd, idx = self._round, len(self._pending)
            self._pending.append(np.array(theta, dtype=np.float64))
            if len(self._pending) >= self._active:
                self._run()
            else:
                while self._round == rnd and self._error is None:
                    self._cond.wait()
            res = self._results.get(rnd)
            if res is None:
                raise RuntimeError("batched evaluation failed") from self._error
            out = (float(res[0][idx]), res[1][idx].copy())
            self._taken[rnd] -= 1
            if self._taken[rnd] == 0:
                del self._results[rnd], self._taken[rnd]
            return out

    def leave(self):
        """this client is done: the others no longer wait for it"""
        with self._cond:
            self._active -= 1
            if self._pending and len(self._pending) >= self._active:
                self._run()
