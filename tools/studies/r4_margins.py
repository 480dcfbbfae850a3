import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import importlib
cv = importlib.import_module("ctrl-vio_amd")
base = [cv.synth.make_window("config1", seed=1300 + i) for i in range(4)]
base[0].knot_const = np.zeros(base[0].K, np.uint8); base[0].knot_const[[0, 1, 2]] = 1
base[1].knot_const = np.zeros(base[1].K, np.uint8); base[1].knot_const[[0, 4, 11, base[1].K - 1]] = 1
base[2].fix_ld = True
base[3].lock_bg = True
def run(n):
    with cv.Solver() as s:
        ws = [base[i % 4].copy() for i in range(n)]
        s.set_windows(ws)
        return ws, s.solve(15)
small, sm_small = run(4)
for rep in range(6):
    big, sm_big = run(208)
    big2, sm_big2 = run(208)
    e_small = max(cv.rel_state_error(big[i], small[i % 4])["state"] for i in range(208))
    e_run = max(cv.rel_state_error(big[i], big2[i])["state"] for i in range(208))
    c_small = max(abs(sm_big[i]["final_cost"] / sm_small[i % 4]["final_cost"] - 1) for i in range(208))
    c_run = max(abs(sm_big[i]["final_cost"] / sm_big2[i]["final_cost"] - 1) for i in range(208))
    its = set((sm_big[i]["iterations"], sm_small[i % 4]["iterations"]) for i in range(208))
    per = [max(cv.rel_state_error(big[i], small[i % 4])["state"] for i in range(j, 208, 4)) for j in range(4)]
    print("state vs small %.2e (per window %s)  run-to-run %.2e | cost vs small %.2e run-to-run %.2e | iters %s" % (e_small, " ".join("%.1e" % x for x in per), e_run, c_small, c_run, its))
