#!/usr/bin/env python3
"""Would RUNS (svdf_k_runs.hip: a row kept in registers across consecutive instances that share it) apply to the uniform rank-pair stream of BASELINE
configs[4]?    The run rule is a file-order fact: instance y may join the run headed at file position h iff every OTHER row it touches
was last touched before h.  This probe plays the rule on a stream with the configs[4] densities scaled by 100 (200 pairs per user, 4 000 touches per item):
user-major runs (the user's row shared) and item-major runs (the positive item's row shared), at most 4 instances per run.  CPU only."""
import numpy as np

rng = np.random.default_rng(1)
NU, NI, PER, R = 10000, 1000, 200, 4
n = NU * PER
u = rng.integers(0, NU, n)
p = rng.integers(0, NI, n)
q = (p + 1 + rng.integers(0, NI - 1, n)) % NI
last_i, last_u = np.full(NI, -1), np.full(NU, -1)
prev_p, prev_q, prev_u = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int64)
for t in range(n):
    prev_p[t], prev_q[t], prev_u[t] = last_i[p[t]], last_i[q[t]], last_u[u[t]]
    last_i[p[t]] = t
    last_i[q[t]] = t
    last_u[u[t]] = t
head, ln, runs = np.full(NU, -1), np.zeros(NU, int), 0
for t in range(n):
    uu = u[t]
    if head[uu] >= 0 and ln[uu] < R and prev_p[t] < head[uu] and prev_q[t] < head[uu]:
        ln[uu] += 1
    else:
        head[uu], ln[uu] = t, 1
        runs += 1
print("user-major runs (<= %d): %.3f pairs per run -> %.3f rows moved per pair (3 without runs)" % (R, n / runs, 2 + runs / n))
head, ln, runs = np.full(NI, -1), np.zeros(NI, int), 0
for t in range(n):
    it = p[t]
    # (the item's own previous touch must be the run's previous member: approximated by "lies inside the run")
    if head[it] >= 0 and ln[it] < R and prev_u[t] < head[it] and prev_q[t] < head[it] and prev_p[t] >= head[it]:
        ln[it] += 1
    else:
        head[it], ln[it] = t, 1
        runs += 1
print("item-major runs on the positive item (<= %d): %.3f pairs per run -> %.3f rows moved per pair" % (R, n / runs, 2 + runs / n))
print("basicMF ratings of configs[1] for comparison (profiles/r05_runs_sweep.txt): 3.0 - 3.5 ratings per run, 2 -> 1.3 rows per rating")
