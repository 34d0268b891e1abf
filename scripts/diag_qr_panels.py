"""Diagnostic: take a dense QR solve that fails the self-check apart, panel by panel, on the host.

For one (state, lambda) of QRKIT on problem-21 (fp64) the factorisation is stopped behind panel p = 0, 1, ... (BA_QR_STOP_PANEL) and the
matrix as the GPU left it (getter 16) and the T factors of that panel (getter 17) are pulled to the host.  numpy then applies panel p's
reflectors -- the GPU's own V (read out of the factored matrix) and the GPU's own T -- to the GPU's state behind panel p - 1 and
compares with what the GPU made of it: a difference in the trailing columns is k_qr_apply's, an inconsistent (V, T) pair is
k_qr_chunk's (checked: T against larft from V, the panel's own columns against R, orthogonality of I - V T V^T).
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BA_DBG_QRCHECK"] = "1"
os.environ["BA_QR_ONE_STREAM"] = "1"
import bundleadjustment_benchmarks_amd as ba

PB, NSB1, NSBU, CH, LEVELS = 32, 16, 16, 512, 8
path = os.path.join(ROOT, "data", "problem-21-11315-pre.txt")
p = ba.Problem.load_bal(path)
D, K, M = p.D, p.K, p.M
m = 2 * K + 3 * M + D
tau_stride = ((m + CH - 1) // CH + 2) * PB * PB
s = ba.Solver(p, ba.QRKIT, ba.F64)
s.linearize(False)


def get(code, n):
    r = np.empty(n)
    ba._chk(ba.lib().ba_solver_get(s._h, code, r.ctypes.data_as(C.c_void_p), n), "get %d" % code)
    return r


def check():
    r = get(14, 2 * D)
    return np.linalg.norm(r[:D]) / np.linalg.norm(r[D:])


# the same lambda sequence as scripts/diag_qrcheck.py, state 0
rng = np.random.default_rng(0)
os.environ.pop("BA_QR_STOP_PANEL", None)
cands = []
for lam0 in (1e-1, 1e-3, 1e-5, 1e-7, 1e-9):
    for rep in range(8):
        lam = lam0 * (1 + 1e-3 * rng.standard_normal())
        s.try_step(lam)
        cands.append((lam, check()))
for lam, v in cands: print("lambda %.15e  self-check %.2e" % (lam, v))
bad = [c for c in cands if c[1] > 1e-10]
good = [c for c in cands if c[1] < 1e-12]
if not bad:
    print("no failing solve among the candidates"); sys.exit(0)


def grow(row0, g, l, stride, nsb_fan):
    return row0 + (g * nsb_fan + (l >> 5)) * stride + (l & 31)


def analyse(lam, tag):
    print("==== %s: lambda %.15e" % (tag, lam))
    os.environ.pop("BA_QR_STOP_PANEL", None)
    s.try_step(lam)
    print("  full solve self-check %.2e" % check())
    A0 = get(15, m * (D + 1)).reshape(D + 1, m).T.copy()  # (m, D + 1)
    prev = A0
    npan = (D + PB - 1) // PB
    for pnl in range(npan):
        os.environ["BA_QR_STOP_PANEL"] = str(pnl)
        s.try_step(lam)
        F = get(16, m * (D + 1)).reshape(D + 1, m).T.copy()
        Tall = get(17, LEVELS * tau_stride)
        c0 = PB * pnl
        bw = min(PB, D - c0)
        col0 = c0 + bw
        X = prev[:, c0:].copy()  # panel columns + trailing columns + rhs, the state the panel starts from
        # columns in front of the panel must not change
        dfront = np.abs(F[:, :c0] - prev[:, :c0]).max() if c0 else 0.0
        nsb = (m - c0 + PB - 1) // PB
        stride = PB
        level = 1
        worst_T = worst_orth = 0.0
        while True:
            fan = NSB1 if level == 1 else NSBU
            nch = (nsb + fan - 1) // fan
            for g in range(nch):
                nsbg = min(fan, nsb - g * fan)
                rows = PB * nsbg
                l = np.arange(rows)
                gr = grow(c0, g, l, stride, fan)
                ok = gr < m + 64
                gr_c = np.minimum(gr, m - 1)
                Fv = np.where((gr < m)[:, None], F[gr_c, c0:c0 + bw], 0.0)
                V = np.zeros((rows, PB))
                cc = np.arange(bw)
                stored = (l[:, None] > cc[None, :])
                if level > 1: stored &= ((l[:, None] >> 5) > 0) & ((l[:, None] & 31) <= cc[None, :])
                V[:, :bw] = np.where(stored, Fv, 0.0)
                V[cc, cc] = 1.0
                Tg = Tall[(level - 1) * tau_stride + g * PB * PB:(level - 1) * tau_stride + (g + 1) * PB * PB].reshape(PB, PB)
                # host larft from V and the GPU's taus (the diagonal of T)
                G = V.T @ V
                tau = np.diag(Tg).copy()
                Th = np.zeros((PB, PB))
                for j in range(PB):
                    Th[j, j] = tau[j]
                    if j: Th[:j, j] = -tau[j] * (Th[:j, :j] @ G[:j, j])
                dT = np.abs(Tg - Th).max() / max(np.abs(Th).max(), 1e-300)
                orth = np.abs(Tg + Tg.T - Tg.T @ G @ Tg).max()
                worst_T = max(worst_T, dT); worst_orth = max(worst_orth, orth)
                if dT > 1e-10 or orth > 1e-10:
                    print("    panel %d level %d chunk %d: |T_gpu - larft(V)| %.2e  |T + T' - T'GT| %.2e  taus %s" % (pnl, level, g, dT, orth, np.array2string(tau[:bw], precision=3)))
                Xg = np.where((gr < m)[:, None], X[gr_c], 0.0)
                Xg = Xg - V @ (Tg.T @ (V.T @ Xg))
                inb = gr < m
                X[gr[inb]] = Xg[inb]
            if nch == 1: break
            nsb = nch
            stride *= fan
            level += 1
        # compare: the panel's own columns -> R on the rows c0 .. c0 + bw - 1 (upper triangle), and the trailing columns + rhs
        Rg = np.triu(F[c0:c0 + bw, c0:c0 + bw])
        Rh = X[c0:c0 + bw, :bw]
        dR = np.abs(np.triu(Rh) - Rg).max() / np.abs(Rg).max()
        low = np.abs(np.tril(Rh, -1)).max() / np.abs(Rg).max()
        # everything of the panel's columns below its R must have been annihilated (rows that carry data of this panel at any level)
        tr_g = F[:, col0:]
        tr_h = X[:, bw:]
        cn = np.linalg.norm(tr_h, axis=0) + 1e-300
        dcol = np.abs(tr_g - tr_h).max(axis=0) / cn
        print("  panel %d: front unchanged %.1e | R: |emul - gpu| %.1e, below-diagonal residue %.1e | T vs larft %.1e, orth %.1e | trailing max col diff %.2e (col %d), rhs diff %.2e"
              % (pnl, dfront, dR, low, worst_T, worst_orth, dcol[:-1].max() if len(dcol) > 1 else 0.0, col0 + int(dcol[:-1].argmax()) if len(dcol) > 1 else -1, dcol[-1]))
        if dcol[-1] > 1e-11 or (len(dcol) > 1 and dcol[:-1].max() > 1e-11):
            j = len(dcol) - 1 if dcol[-1] > 1e-11 else int(dcol[:-1].argmax())
            d = np.abs(tr_g[:, j] - tr_h[:, j])
            rowsbad = np.nonzero(d > 1e-3 * d.max())[0]
            print("    column %d: %d rows differ; first %s ... last %s; max |diff| %.3e at row %d (gpu %.6e, emul %.6e)" % (col0 + j, len(rowsbad), rowsbad[:12], rowsbad[-6:], d.max(), d.argmax(), tr_g[d.argmax(), j], tr_h[d.argmax(), j]))
        # Gram invariant of the GPU's own state: columns behind the panel keep their inner products
        Gp = prev[:, col0:].T @ prev[:, col0:]
        Gn = F[:, col0:].T @ F[:, col0:]
        dn = np.sqrt(np.outer(np.diag(Gp), np.diag(Gp))) + 1e-300
        gd = np.abs(Gn - Gp) / dn
        print("    Gram drift of the GPU's trailing block: matrix columns %.2e, against rhs %.2e" % (gd[:-1, :-1].max() if gd.shape[0] > 1 else 0.0, gd[-1].max()))
        prev = F
    os.environ.pop("BA_QR_STOP_PANEL", None)


analyse(bad[0][0], "FAILING")
if good: analyse(good[0][0], "GOOD (control)")
