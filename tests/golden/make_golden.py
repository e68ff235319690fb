#!/usr/bin/env python3
"""Generates tests/golden/*.npz: high-precision known answers for the pieces of the hot path
whose mathematics the reference states explicitly (acados/mpcPathTracking.py:41-47 model,
:63-80 cost, :103 W; config/mpc_params.yaml:41,59 dt and wheelbase).

The reference holds no golden vectors and acados/CasADi are absent, so these
pin the CPU oracle against an INDEPENDENT formulation:
  * rk4_map.npz   -- the classical RK4 map phi(x,u;p) evaluated with mpmath at 50 digits and its
                     Jacobians A = dphi/dx, B = dphi/du by 50-digit central differences of that
                     map (the oracle integrates the variational ODE instead);
  * nls_cost.npz  -- y, J = dy/d[x;u], H = dt J'WJ, g = dt J'W y from sympy-differentiated
                     closed forms evaluated at 50 digits.
Run:  python tests/golden/make_golden.py   (deterministic; needs mpmath and sympy)
"""
import os

import mpmath as mp
import numpy as np
import sympy as sy

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50
L = mp.mpf("2.65")          # config/mpc_params.yaml:59
DT = mp.mpf("0.35")         # config/mpc_params.yaml:41
W = [5, 5, 10, mp.mpf("0.01"), mp.mpf("0.1"), mp.mpf("0.5"), mp.mpf("0.1")]   # mpcPathTracking.py:103


def f(x, u, p):
    """mpcPathTracking.py:41-47"""
    psi, v, delta, thr = x[2], x[3], x[4], x[5]
    return [v * mp.cos(psi), v * mp.sin(psi), v / L * mp.tan(delta),
            5 * thr - mp.mpf("0.087") * v + mp.sin(p[4]) * mp.mpf("9.81"), u[0], u[1]]


def rk4(x, u, p, h):
    k1 = f(x, u, p)
    k2 = f([a + h / 2 * b for a, b in zip(x, k1)], u, p)
    k3 = f([a + h / 2 * b for a, b in zip(x, k2)], u, p)
    k4 = f([a + h * b for a, b in zip(x, k3)], u, p)
    return [a + h / 6 * (b + 2 * c + 2 * d + e) for a, b, c, d, e in zip(x, k1, k2, k3, k4)]


def jac_fd(x, u, p, h):
    eps = mp.mpf(10) ** -20
    z = list(x) + list(u)
    J = mp.zeros(6, 8)
    for j in range(8):
        zp, zm = list(z), list(z)
        zp[j] += eps
        zm[j] -= eps
        fp = rk4(zp[:6], zp[6:], p, h)
        fm = rk4(zm[:6], zm[6:], p, h)
        for i in range(6):
            J[i, j] = (fp[i] - fm[i]) / (2 * eps)
    return J


def sample(rng, n):
    x = np.zeros((n, 6)); u = np.zeros((n, 2)); p = np.zeros((n, 6))
    x[:, 0] = rng.uniform(-5, 30, n); x[:, 1] = rng.uniform(-5, 5, n)
    x[:, 2] = rng.uniform(-1.5, 1.5, n); x[:, 3] = rng.uniform(-1, 9, n)
    x[:, 4] = rng.uniform(-0.57, 0.57, n); x[:, 5] = rng.uniform(-1, 1, n)
    u[:, 0] = rng.uniform(-2, 2, n); u[:, 1] = rng.uniform(-0.33, 0.33, n)
    p[:, 0] = rng.normal(0, 0.5, n); p[:, 1] = rng.uniform(-1.5, 1.5, n)
    p[:, 2] = rng.uniform(-0.1, 0.1, n); p[:, 3] = rng.uniform(-0.002, 0.002, n)
    p[:, 4] = rng.uniform(-0.05, 0.05, n); p[:, 5] = rng.uniform(3, 7, n)
    return x, u, p


def main():
    rng = np.random.default_rng(0x5EED0001)
    n = 64
    x, u, p = sample(rng, n)
    phi = np.zeros((n, 6)); A = np.zeros((n, 6, 6)); B = np.zeros((n, 6, 2))
    for s in range(n):
        xs = [mp.mpf(float(v)) for v in x[s]]; us = [mp.mpf(float(v)) for v in u[s]]
        ps = [mp.mpf(float(v)) for v in p[s]]
        ph = rk4(xs, us, ps, DT)
        J = jac_fd(xs, us, ps, DT)
        phi[s] = [float(v) for v in ph]
        for i in range(6):
            for j in range(8):
                if j < 6:
                    A[s, i, j] = float(J[i, j])
                else:
                    B[s, i, j - 6] = float(J[i, j])
    np.savez(os.path.join(HERE, "rk4_map.npz"), x=x, u=u, p=p, dt=float(DT), wheelbase=float(L),
             phi=phi, A=A, B=B)

    # NLS cost (mpcPathTracking.py:63-80): symbolic Jacobian
    X, Y, PSI, V, D, T, U0, U1 = sy.symbols("X Y PSI V D T U0 U1", real=True)
    c0, c1, c2, c3, vref = sy.symbols("c0 c1 c2 c3 vref", real=True)
    path = c3 * X ** 3 + c2 * X ** 2 + c1 * X + c0
    yexpr = sy.Matrix([path - Y, PSI - sy.atan(3 * c3 * X ** 2 + 2 * c2 * X + c1), V - vref, D, T, U0, U1])
    zsym = sy.Matrix([X, Y, PSI, V, D, T, U0, U1])
    Jexpr = yexpr.jacobian(zsym)
    args = (X, Y, PSI, V, D, T, U0, U1, c0, c1, c2, c3, vref)
    yfun = sy.lambdify(args, yexpr, "mpmath")
    Jfun = sy.lambdify(args, Jexpr, "mpmath")
    x, u, p = sample(rng, n)
    ys = np.zeros((n, 7)); Js = np.zeros((n, 7, 8)); Hs = np.zeros((n, 8, 8)); gs = np.zeros((n, 8))
    Wm = mp.diag(W)
    for s in range(n):
        a = [mp.mpf(float(v)) for v in list(x[s]) + list(u[s]) + list(p[s, :4]) + [p[s, 5]]]
        yv = yfun(*a); Jv = Jfun(*a)
        Hv = DT * (Jv.T * Wm * Jv); gv = DT * (Jv.T * Wm * yv)
        ys[s] = [float(v) for v in yv]
        Js[s] = np.array([[float(Jv[i, j]) for j in range(8)] for i in range(7)])
        Hs[s] = np.array([[float(Hv[i, j]) for j in range(8)] for i in range(8)])
        gs[s] = [float(v) for v in gv]
    np.savez(os.path.join(HERE, "nls_cost.npz"), x=x, u=u, p=p, dt=float(DT), W=np.array([float(w) for w in W]),
             y=ys, J=Js, H=Hs, g=gs)
    print("wrote rk4_map.npz, nls_cost.npz")


if __name__ == "__main__":
    main()
