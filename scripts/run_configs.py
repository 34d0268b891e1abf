"""Runs the BASELINE.json configs to termination on the GPU (and the oracle where it finishes in about a minute):
final energy, status, trials, trials/s, Schur-solve ms.  Usage: python scripts/run_configs.py [cfg ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bundleadjustment_benchmarks_amd as ba
import oracle_lib as O
from bench import WORKLOADS, load_problem
from conftest import to_oracle

cfgs = sys.argv[1:] or ["cfg2", "cfg3", "cfg1", "cfg4"]
rows = []


def _heartbeat():  # (the oracle's free run at config 4 is minutes of silence: gpurun takes a silent command for hung)
    import threading
    def beat():
        while True:
            time.sleep(60)
            print("... running (%s)" % time.strftime("%H:%M:%S"), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


_heartbeat()
for name in cfgs:
    kind_s, scalar_s, _, _, _ = WORKLOADS[name]
    kind = {"QRKIT": ba.QRKIT, "QRCHOL": ba.QRCHOL, "CHOLESKY": ba.CHOLESKY}[kind_s]
    scalar = ba.F64 if scalar_s == "f64" else ba.F32
    prob, source = load_problem(ba, name)
    s = ba.Solver(prob, kind, scalar)
    st0 = s.stats()
    s.minimize(max_trials=3, trace=False)  # warm-up
    s = ba.Solver(prob, kind, scalar)
    cap = 3000
    t0 = time.perf_counter()
    r = s.minimize(max_trials=cap, trace=True)
    el = time.perf_counter() - t0
    st1 = s.stats()
    row = {"config": name, "solver": kind_s, "dtype": scalar_s, "source": source, "N": prob.N, "M": prob.M, "K": prob.K,
           "gpu_status": ba.STATUS[r["status"]], "gpu_trials": r["trials"], "gpu_iterations": r["iterations"], "gpu_seconds": el,
           "gpu_trials_per_s": r["trials"] / el, "gpu_schur_ms": r["schur_ms"], "gpu_final_energy": r["energy"],
           "initial_energy": float(r["trace"][0, 2]), "mean_err_before": st0["mean_err"], "mean_err_after": st1["mean_err"],
           "inliers_before": st0["n_inliers"], "inliers_after": st1["n_inliers"]}
    if name in ("cfg2", "cfg3", "cfg1", "cfg4"):
        po = to_oracle(prob)
        if name == "cfg4":
            O.set_threads(os.cpu_count() or 1)  # (328 trials at D = 2313: minutes on one core; the results do not depend on the thread count)
        okind = {"QRKIT": O.QRCHOL, "QRCHOL": O.QRCHOL, "CHOLESKY": O.CHOLESKY}[kind_s]
        t0 = time.perf_counter()
        ro = O.minimize(okind, po, dtype=np.float64 if scalar_s == "f64" else np.float32, max_trials=cap)
        elo = time.perf_counter() - t0
        tr = ro["trace"]
        e_fin = tr[-1, 6] if tr[-1, 1] == 1 else tr[-1, 2]
        n = min(len(tr), len(r["trace"]))
        same = np.where(tr[:n, 1] != r["trace"][:n, 1])[0]
        dto = np.float64 if scalar_s == "f64" else np.float32
        so0 = O.stats(po, O.init_cams(po, dto), po.pts.astype(dto))
        so1 = O.stats(po, ro["cam15"], ro["pts"])
        row.update({"cpu_mean_err_before": so0["mean_err"], "cpu_mean_err_after": so1["mean_err"], "cpu_inliers_after": so1["n_inliers"],
                    "cpu_objective_after": so1["objective"], "gpu_objective_after": st1["objective"]})
        if not np.isfinite(st1["mean_err"]) or not np.isfinite(so1["mean_err"]):
            # which observation goes non-finite, on which side, and why (fp64 re-evaluation of the final parameters)
            def worst(cam15, pts, tag):
                c = np.asarray(cam15, np.float64).reshape(-1, 15)
                X = np.asarray(pts, np.float64).reshape(-1, 3)
                ci, pi = po.cam_idx, po.pt_idx
                R = c[ci, :9].reshape(-1, 3, 3)
                XX = np.einsum("kij,kj->ki", R, X[pi]) + c[ci, 9:12]
                xu = XX[:, :2] / XX[:, 2:3]
                r2 = (xu ** 2).sum(1)
                kr = 1 + c[ci, 13] * r2 + c[ci, 14] * r2 * r2
                q = c[ci, 12:13] * kr[:, None] * xu
                err = np.linalg.norm(q - po.meas.reshape(-1, 2), axis=1)
                with np.errstate(over="ignore", invalid="ignore"):
                    r2f = r2.astype(np.float32)
                    ovf = ~np.isfinite((r2f * r2f * c[ci, 14].astype(np.float32)))
                k = int(np.nanargmax(np.where(np.isfinite(err), err, np.inf)))
                return {tag + "_worst_obs": k, tag + "_worst_err_fp64": float(err[k]), tag + "_worst_depth_XX2": float(XX[k, 2]),
                        tag + "_worst_r2u": float(r2[k]), tag + "_nonfinite_in_fp64": int((~np.isfinite(err)).sum()),
                        tag + "_float_overflow_of_k2_r4": int(ovf.sum()), tag + "_n_err_gt_1e6": int((err > 1e6).sum())}
            row.update(worst(s.get(ba.GET_CAMS), s.get(ba.GET_POINTS), "gpu"))
            row.update(worst(ro["cam15"], ro["pts"], "cpu"))
        row.update({"cpu_status": ba.STATUS[ro["status"]], "cpu_trials": len(tr), "cpu_seconds": elo, "cpu_trials_per_s": len(tr) / elo,
                    "cpu_final_energy": float(e_fin), "rel_final_energy_diff": abs(float(e_fin) - r["energy"]) / float(e_fin),
                    "first_accept_reject_difference_at_trial": int(same[0]) + 1 if len(same) else None})
    # what "the final cost" resolves to on this input: the fp64 oracle's own free runs from inputs perturbed by 1e-13, and the free run
    # in quad precision (tests/golden/make_referee.py; problem-21 only)
    gold = os.path.join(ROOT, "tests", "golden")
    tag = {"cfg2": "problem21_qrchol"}.get(name)
    if tag:
        ens = json.load(open(os.path.join(gold, "referee_ensemble_%s.json" % tag)))
        fr = json.load(open(os.path.join(gold, "referee_freerun_%s.json" % tag)))
        row.update({"oracle_ensemble_1e-13_final_energy_min": ens["final_energy_min"], "oracle_ensemble_1e-13_final_energy_max": ens["final_energy_max"],
                    "oracle_ensemble_1e-13_final_energy_median": ens["final_energy_median"], "quad_free_run_final_energy": fr["final_energy_quad"],
                    "quad_free_run_trials": fr["trials"],
                    "abs_diff_gpu_vs_quad_final": abs(r["energy"] - fr["final_energy_quad"]),
                    "abs_diff_cpu_vs_quad_final": abs(row["cpu_final_energy"] - fr["final_energy_quad"]),
                    "gpu_final_inside_ensemble_range": bool(ens["final_energy_min"] <= r["energy"] <= ens["final_energy_max"])})
    rows.append(row)
    print(json.dumps(row), flush=True)
