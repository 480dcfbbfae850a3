"""CPU: a NumPy model of the control flow of k_cholesky_solve (csrc/kernels_solve.hpp) -- 32-column panels, 16-row tiles that take part in
a panel iff env_first[R] <= jb / 16 + 1, trailing updates over the pairs of tiles that take part, block back-substitution that reaches back
to each tile's own start -- run on the reduced system of real windows with every entry OUTSIDE the planned envelope set to NaN.  The model
reads exactly what the kernel reads: a NaN in the solution means the kernel would read a tile nobody wrote.  The solution must equal the
dense solve."""
import numpy as np
import pytest

from test_sparsity_plan import hp, plan  # noqa: F401  (fixture + helper)


def panel_cholesky_model(S, rhs, ef):
    """S: (P, P) lower triangle used (NaN outside the envelope), rhs (P,), ef per 16-row tile (P // 16 + 1 entries).  Returns x, tile products."""
    S = S.copy(); y = rhs.copy()
    P = S.shape[0]
    prods = 0
    Linv_blocks = {}
    for jb in range(0, P, 32):
        nb = min(32, P - jb); r0 = jb + nb; nt = P - r0; ntr = nt + 1
        RS = (ntr + 15) // 16 * 16; ntile = RS // 16
        R0 = r0 >> 4
        plist = [t for t in range(ntile) if ef[min(R0 + t, P // 16)] <= (jb >> 4) + 1]
        A11 = np.tril(S[jb:jb + nb, jb:jb + nb])
        A11 = A11 + np.tril(A11, -1).T
        L11 = np.linalg.cholesky(A11)
        Linv = np.linalg.inv(L11)
        Linv_blocks[jb // 32] = Linv
        panel = {}                                   # local row -> L21 row
        for t in plist:
            for r in range(16 * t, 16 * t + 16):
                if r >= ntr:
                    continue
                src = S[r0 + r, jb:jb + nb] if r < nt else y[jb:jb + nb]
                l21 = src @ Linv.T
                panel[r] = l21
                if r < nt:
                    S[r0 + r, jb:jb + nb] = l21
                else:
                    y[jb:jb + nb] = l21
        if nt > 0:
            assert plist[:2] == [0, 1][:min(2, ntile)], (jb, plist)      # the next diagonal block takes part (look-ahead)
            for a in range(len(plist)):
                for b in range(a + 1):
                    ti, tj = plist[a], plist[b]
                    prods += 2
                    for row in range(16 * ti, 16 * ti + 16):
                        for col in range(16 * tj, 16 * tj + 16):
                            if col < nt and ((row < nt and col <= row) or row == nt):
                                v = panel[row] @ panel[col]
                                if row < nt:
                                    S[r0 + row, r0 + col] -= v
                                else:
                                    y[r0 + col] -= v
    xs = y.copy()
    nblk = (P + 31) // 32
    for b in range(nblk - 1, -1, -1):
        jb = 32 * b; nb = min(32, P - jb)
        ca = 16 * ef[2 * b]; cb = 16 * ef[min(2 * b + 1, P // 16)]
        xb = Linv_blocks[b].T @ xs[jb:jb + nb]
        xs[jb:jb + nb] = xb
        for j in range(min(ca, cb), jb):
            s = 0.0
            for ii in range(nb):
                if j >= (ca if ii < 16 else cb):
                    s += S[jb + ii, j] * xb[ii]
            xs[j] -= s
    return xs, prods


@pytest.mark.parametrize("cfg,kw", [("config1", dict(F=16, L=60, M=750)), ("config1", dict(F=20, L=40, M=900)), ("tiny", dict(F=14, L=30, M=400))])
def test_envelope_panel_cholesky_reads_only_what_is_written(hp, cv, oracle, cfg, kw):
    w = cv.synth.make_window(cfg, seed=1400, **kw)
    P = w.P
    assert P > 223                                   # the panel kernel's territory
    pl = plan(hp, cv, w)
    ef = pl["env"]
    H, g, cost = oracle.OracleWindow(w.copy()).build_normal()
    Hpp, W, Hll = H[:P, :P], H[:P, P:], np.diag(H)[P:]
    D = 1e-4 * np.diag(Hpp) + 1e-6
    dl = 1e-4 * Hll + 1e-6
    S = Hpp + np.diag(D) - (W / (Hll + dl)) @ W.T
    rhs = -g[:P] + (W / (Hll + dl)) @ g[P:]
    Sm = np.full((P, P), np.nan)
    for i in range(P):
        c0 = 16 * ef[i // 16]
        Sm[i, c0:i + 1] = S[i, c0:i + 1]
        assert np.all(S[i, :c0] == 0.0)              # (what the plan drops is structurally zero)
    x, prods = panel_cholesky_model(Sm, rhs, ef)
    assert np.all(np.isfinite(x))
    xd = np.linalg.solve(S, rhs)
    assert np.abs(x - xd).max() <= 1e-9 * np.abs(xd).max()
    nt = P // 16 + 1
    dense = sum(2 * n * (n + 1) // 2 for n in [((P - min(jb + 32, P) + 1 + 15) // 16) for jb in range(0, P, 32)] if n > 0)
    assert prods < dense
