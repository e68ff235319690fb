"""The oracle's OCP-QP solver and SQP-RTI step, certified without acados:
  * KKT certificate of the returned QP solution (strictly convex QP => unique minimiser);
  * an independent dense solve: condense nothing, assemble the full KKT system of the
    equality-constrained QP on the oracle's final active set with numpy and solve it;
  * properties of the SQP-RTI step the reference's protocol implies.
"""
import numpy as np
import pytest

from mpc_local_planner_amd import scenarios as sc
from oracle import oracle as orc

NX, NU, NZ = 6, 2, 8


def _linearise(cfg, x0, p, xt, ut):
    N = cfg.N
    A = np.zeros((N, 6, 6)); B = np.zeros((N, 6, 2)); b = np.zeros((N, 6))
    H = np.zeros((N, 8, 8)); g = np.zeros((N, 8))
    lu = np.zeros((N, 2)); uu = np.zeros((N, 2)); lx = np.zeros((N, 2)); ux = np.zeros((N, 2))
    for k in range(N):
        phi, A[k], B[k] = orc.rk4_sens(cfg, xt[k], ut[k], p)
        b[k] = phi - xt[k + 1]
        H[k], g[k] = orc.stage_gn(cfg, xt[k], ut[k], p)
        lu[k] = np.array(cfg.lbu[:]) - ut[k]; uu[k] = np.array(cfg.ubu[:]) - ut[k]
        lx[k] = np.array(cfg.lbx[:]) - xt[k, 4:6]; ux[k] = np.array(cfg.ubx[:]) - xt[k, 4:6]
    return A, B, b, H, g, lu, uu, lx, ux, x0 - xt[0]


def _dense_active_set_solve(A, B, b, H, g, lu, uu, lx, ux, dx0, act_lo, act_up):
    """Full KKT system in (dx_1..dx_N, du_0..du_{N-1}) with the dynamics and the active bounds
    as equalities; returns dx[N+1,6], du[N,2]."""
    N = A.shape[0]
    nv = N * 6 + N * 2
    ix = lambda k: slice((k - 1) * 6, k * 6)          # dx_k, k >= 1
    iu = lambda k: slice(N * 6 + k * 2, N * 6 + k * 2 + 2)
    Q = np.zeros((nv, nv)); q = np.zeros(nv)
    for k in range(N):
        Hxx, Hxu, Huu = H[k][:6, :6], H[k][:6, 6:], H[k][6:, 6:]
        Q[iu(k), iu(k)] += Huu
        q[iu(k)] += g[k][6:]
        if k == 0:
            q[iu(0)] += Hxu.T @ dx0
        else:
            Q[ix(k), ix(k)] += Hxx
            Q[ix(k), iu(k)] += Hxu
            Q[iu(k), ix(k)] += Hxu.T
            q[ix(k)] += g[k][:6]
    rows, rhs = [], []
    for k in range(N):          # dx_{k+1} - A dx_k - B du_k = b_k
        for i in range(6):
            r = np.zeros(nv)
            r[(k) * 6 + i] = 1.0
            if k >= 1:
                r[ix(k)] -= A[k][i]
            r[iu(k)] -= B[k][i]
            rows.append(r)
            rhs.append(b[k][i] + (A[k][i] @ dx0 if k == 0 else 0.0))
    for k in range(N):
        for j in range(4):
            lo = lu[k][j] if j < 2 else lx[k][j - 2]
            up = uu[k][j] if j < 2 else ux[k][j - 2]
            if j >= 2 and k == 0:
                continue
            col = N * 6 + k * 2 + j if j < 2 else (k - 1) * 6 + 4 + (j - 2)
            for flag, val in ((act_lo[k, j], lo), (act_up[k, j], up)):
                if flag:
                    r = np.zeros(nv); r[col] = 1.0
                    rows.append(r); rhs.append(val)
    C = np.array(rows); d = np.array(rhs)
    KKT = np.block([[Q, C.T], [C, np.zeros((C.shape[0], C.shape[0]))]])
    sol = np.linalg.solve(KKT, np.concatenate([-q, d]))
    dx = np.zeros((N + 1, 6)); du = np.zeros((N, 2))
    dx[0] = dx0
    dx[1:] = sol[:N * 6].reshape(N, 6)
    du[:] = sol[N * 6:nv].reshape(N, 2)
    return dx, du


@pytest.mark.parametrize("gen,N", [(sc.config2, 12), (sc.config4, 12), (sc.config2, 40), (sc.config4, 25)])
def test_qp_kkt_certificate_and_dense_check(gen, N):
    cfg = orc.default_config(N, 0.35)
    x0s, ps = gen(12)
    n_active = 0
    for i in range(x0s.shape[0]):
        xt, ut = orc.init_guess(cfg, x0s[i])
        if i % 2:   # a warm, non-trivial linearisation point
            orc.rti(cfg, x0s[i], ps[i], xt, ut)
        A, B, b, H, g, lu, uu, lx, ux, dx0 = _linearise(cfg, x0s[i], ps[i], xt, ut)
        r = orc.qp_ipm(cfg, A, B, b, H, g, lu, uu, lx, ux, dx0)
        assert r["status"] == 0
        dx, du, lam, pi = r["dx"], r["du"], r["lam"], r["pi"]
        scale = max(1.0, np.abs(dx).max(), np.abs(du).max())
        # primal: dynamics and bounds
        assert np.abs(dx[0] - dx0).max() == 0
        for k in range(N):
            assert np.abs(A[k] @ dx[k] + B[k] @ du[k] + b[k] - dx[k + 1]).max() <= 1e-10 * scale
        assert (du >= lu - 1e-10).all() and (du <= uu + 1e-10).all()
        assert (dx[1:N, 4:6] >= lx[1:] - 1e-10).all() and (dx[1:N, 4:6] <= ux[1:] + 1e-10).all()
        # dual feasibility, complementarity
        assert (lam >= 0).all()
        wl = np.concatenate([du - lu, dx[:N, 4:6] - lx], axis=1)
        wu = np.concatenate([uu - du, ux - dx[:N, 4:6]], axis=1)
        wl[0, 2:] = 0; wu[0, 2:] = 0
        assert np.abs(lam[:, :4] * wl).max() <= 1e-9 and np.abs(lam[:, 4:] * wu).max() <= 1e-9
        # stationarity with the returned costates
        for k in range(N):
            z = np.concatenate([dx[k], du[k]])
            grad = H[k] @ z + g[k]
            ru = grad[6:] + B[k].T @ pi[k + 1] - lam[k, 0:2] + lam[k, 4:6]
            assert np.abs(ru).max() <= 1e-8 * max(1.0, np.abs(grad).max()), (k, ru)
            if k >= 1:
                rx = grad[:6] + A[k].T @ pi[k + 1] - pi[k]
                rx[4:6] += -lam[k, 2:4] + lam[k, 6:8]
                assert np.abs(rx).max() <= 1e-8 * max(1.0, np.abs(grad).max()), (k, rx)
        # independent dense solve on the identified active set
        act_lo = lam[:, :4] > 1e-7
        act_up = lam[:, 4:] > 1e-7
        n_active += act_lo.sum() + act_up.sum()
        dxd, dud = _dense_active_set_solve(A, B, b, H, g, lu, uu, lx, ux, dx0, act_lo, act_up)
        assert np.abs(dxd - dx).max() <= 1e-7 * scale, np.abs(dxd - dx).max()
        assert np.abs(dud - du).max() <= 1e-7 * scale, np.abs(dud - du).max()
    assert n_active > 0      # the cases do exercise active bounds


def test_rti_protocol_properties():
    N = 40
    cfg = orc.default_config(N, 0.35)
    x0s, ps = sc.config2(16)
    for i in range(16):
        xt, ut = orc.init_guess(cfg, x0s[i])
        assert np.array_equal(xt, np.tile(x0s[i], (N + 1, 1))) and not ut.any()   # AcadosSolver.cpp:78-92
        st = orc.rti(cfg, x0s[i], ps[i], xt, ut)
        assert st.status == 0 and st.sqp_iters == 1
        assert np.array_equal(xt[0], x0s[i])             # lbx0 = ubx0 = x0 (AcadosSolver.cpp:15-26)
        assert ut[N - 1].tolist() == [0.0, 0.0] or np.abs(ut[N - 1]).max() < 1e-12   # no terminal cost
        c = cfg
        assert (ut[:, 0] >= c.lbu[0] - 1e-9).all() and (ut[:, 1] <= c.ubu[1] + 1e-9).all()
        # mirror symmetry of the OCP
        xm0 = x0s[i].copy(); pm = ps[i].copy()
        xm0[2] *= -1; xm0[4] *= -1; pm[:4] *= -1
        xm, um = orc.init_guess(cfg, xm0)
        orc.rti(cfg, xm0, pm, xm, um)
        assert np.abs(um[:, 0] + ut[:, 0]).max() < 1e-6 and np.abs(um[:, 1] - ut[:, 1]).max() < 1e-6
        assert np.abs(xm[:, 1] + xt[:, 1]).max() < 1e-6 and np.abs(xm[:, 3] - xt[:, 3]).max() < 1e-6


def test_sqp_fixed_point_and_warm_resolve():
    N = 30
    cfg = orc.default_config(N, 0.35)
    x0s, ps = sc.config2(6)
    for i in range(6):
        xt, ut = orc.init_guess(cfg, x0s[i])
        st = orc.sqp(cfg, x0s[i], ps[i], xt, ut, 30, 1e-8)
        assert st.status == 0, (st.status, st.sqp_iters, st.res_stat, st.res_eq)
        assert max(st.res_stat, st.res_eq, st.res_ineq, st.res_comp) <= 1e-8
        # an RTI call from the converged iterate does not move it
        x2, u2 = xt.copy(), ut.copy()
        st2 = orc.rti(cfg, x0s[i], ps[i], x2, u2)
        assert st2.status == 0 and st2.step_inf <= 1e-6
