"""An INDEPENDENT known answer for the rod's internal force / torque balance in the geometrically
nonlinear regime (TEST INFRASTRUCTURE): the static equilibrium of the discrete Cosserat rod of
Gazzola et al. 2018 — n straight elements with rest length l, bending hinges at the n - 1 interior
nodes, shear and stretch strains sigma = e Q t - z, stress n = S sigma, lab-frame force Q^T n / e,
couples B kappa / eps^3 — clamped at its first element and loaded by a dead tip force F, solved
DIRECTLY as a small nonlinear system (scipy fsolve) instead of by time stepping:

    in-plane, element j at director angle t_j (t_0 = 0), local force components (F sin t_j, F cos t_j):
      sigma_3 = e F sin t / EA,  sigma_1 = e F cos t / (alpha_c G A),  e = |(1 + sigma_3, sigma_1)|
      edge_j  = l [(1 + sigma_3) d3 + sigma_1 d1']                        (d3 = (cos t, sin t), d1' = (-sin t, cos t))
      hinge k: EI (t_k - t_{k-1}) / (l eps_k^3) = F (x_tip - x_k),         eps_k = (e_k + e_{k-1}) / 2

No time integrator, no damper, no rotation update is involved, so agreement with a stepper's
settled or fixed-point state checks its sigma, S, Q^T n / e, kappa = -log(Q+ Q^T) / D (the
theta / sin theta factor at finite joint angles), the eps^3 on the couple and the shear couple's
lever arm at once.  The C oracle settles onto this solution to 1e-11 of the tip position."""
import numpy as np
from scipy.optimize import fsolve


def solve(n, F, EI, GA, EA, L=1.0):
    """-> (theta[n], nodes x[2, n+1]) of the clamped rod under the tip force (0, F)."""
    l = L / n

    def edges(t):
        e = np.ones_like(t)
        for _ in range(8):                      # sigma = e Q F / S with e = |sigma + z|: a contraction (e - 1 ~ 1e-4)
            s3 = e * F * np.sin(t) / EA
            s1 = e * F * np.cos(t) / GA
            e = np.sqrt((1 + s3) ** 2 + s1 ** 2)
        ex = l * ((1 + s3) * np.cos(t) - s1 * np.sin(t))
        ey = l * ((1 + s3) * np.sin(t) + s1 * np.cos(t))
        return ex, ey, e

    def residual(th):
        t = np.concatenate([[0.0], th])
        ex, _, e = edges(t)
        arm = np.cumsum(ex[::-1])[::-1]         # x_tip - x_k for k = 0 .. n-1
        eps = 0.5 * (e[1:] + e[:-1])
        return EI * (t[1:] - t[:-1]) / l / eps ** 3 - F * arm[1:]

    th = fsolve(residual, np.linspace(0.02, 0.9, n - 1), xtol=1e-14, full_output=False)
    assert np.abs(residual(th)).max() < 1e-10 * max(F, 1e-30)
    t = np.concatenate([[0.0], th])
    ex, ey, _ = edges(t)
    x = np.zeros((2, n + 1))
    x[0, 1:], x[1, 1:] = np.cumsum(ex), np.cumsum(ey)
    return t, x


def state(n, F, EI, GA, EA, L=1.0, phi=0.0):
    """The equilibrium as a rod state in the frame of reset_straight(direction +x, normal +z):
    x (3, n+1), Q (3, 3, n) with rows d1, d2 = d3 x d1, d3, for the tip force F (0, cos phi, sin phi).
    phi = 0: bending in the x-y plane about d1 = z.  phi != 0: the same planar solution turned about
    the rod's rest axis — the cross-section is circular, so it is an equilibrium too — with the frames
    carried along WITHOUT twist from the clamped base frame (every element's frame is the base frame
    turned about the fixed bending axis x cross f), so that the curvature has components on BOTH d1
    and d2: the 3-D form of the same check."""
    t, x2 = solve(n, F, EI, GA, EA, L)
    c, s = np.cos(phi), np.sin(phi)
    x = np.zeros((3, n + 1))
    x[0], x[1], x[2] = x2[0], c * x2[1], s * x2[1]
    b = np.array([0.0, -s, c])                              # bending axis x_hat x f_hat
    K = np.array([[0.0, -b[2], b[1]], [b[2], 0.0, -b[0]], [-b[1], b[0], 0.0]])
    base = np.array([[0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 0.0, 0.0]])      # rows d1, d2, d3 of reset_straight
    Q = np.zeros((3, 3, n))
    for k in range(n):
        R = np.eye(3) + np.sin(t[k]) * K + (1 - np.cos(t[k])) * (K @ K)      # rotation about b by t_k
        Q[:, :, k] = base @ R.T                                                # each director (a row) turned
    return t, x, Q


def bending_modes(n, EI, GA, rhoA, rhoI, L=1.0):
    """Small planar vibrations of the clamped discrete rod about the straight state, as a generalised
    eigenproblem assembled here from the discrete energies (nothing of a stepper involved):
        V = 1/2 sum_k GA l (y'_k - theta_k)^2 + 1/2 sum_k EI (theta_{k+1} - theta_k)^2 / l,  y'_k = (y_{k+1} - y_k) / l
        T = 1/2 sum_k m_k ydot_k^2 + 1/2 sum_k (rhoI l) thetadot_k^2,   m_k = rhoA l (half at both ends)
    with y_0 = theta_0 = 0.  -> (omega[...], modes as (y[n+1], theta[n]) columns, mass matrix diag)."""
    from scipy.linalg import eigh

    l = L / n
    ny, nd = n + 1, 2 * n + 1
    K = np.zeros((nd, nd))
    for k in range(n):
        g = np.zeros(nd)
        g[k + 1], g[k], g[ny + k] = 1 / l, -1 / l, -1.0
        K += GA * l * np.outer(g, g)
    for k in range(n - 1):
        b = np.zeros(nd)
        b[ny + k + 1], b[ny + k] = 1.0, -1.0
        K += EI / l * np.outer(b, b)
    m = np.full(ny, rhoA * l)
    m[[0, -1]] *= 0.5
    Md = np.concatenate([m, np.full(n, rhoI * l)])
    free = [i for i in range(nd) if i not in (0, ny)]
    w2, V = eigh(K[np.ix_(free, free)], np.diag(Md[free]))
    modes = np.zeros((nd, len(free)))
    modes[free] = V
    return np.sqrt(w2), modes, Md


def frequency_from_samples(q, h, dt):
    """omega of a pure cosine sampled every h, from q(t+h) + q(t-h) = 2 cos(w_d h) q(t), corrected for the
    position-Verlet step dt: sin(w_d dt / 2) = w dt / 2."""
    q = np.asarray(q)
    c = (q[2:] + q[:-2]) / (2 * q[1:-1])
    sel = np.abs(q[1:-1]) > 0.3 * np.abs(q).max()
    wd = np.arccos(np.clip(c[sel], -1.0, 1.0)).mean() / h
    return 2.0 / dt * np.sin(wd * dt / 2), float(np.std(c[sel]))
