"""Writes expected.json: closed-form K / f_int of sm01's first load step."""
import json
import numpy as np

E, nu = 1e4, 0.3
a = E / (1 - nu * nu); G = a * (1 - nu) / 2
D = np.array([[a, a * nu, 0], [a * nu, a, 0], [0, 0, G]])
X = np.array([[0., 0.], [1., 0.], [1., 1.], [0., 1.]])
K = np.zeros((8, 8))
for tri in ([0, 1, 2], [0, 2, 3]):
    x = X[tri]
    A2 = np.linalg.det(np.c_[np.ones(3), x])
    b = np.array([x[1, 1] - x[2, 1], x[2, 1] - x[0, 1], x[0, 1] - x[1, 1]]) / A2
    c = np.array([x[2, 0] - x[1, 0], x[0, 0] - x[2, 0], x[1, 0] - x[0, 0]]) / A2
    B = np.zeros((3, 6)); B[0, 0::2] = b; B[1, 1::2] = c; B[2, 0::2] = c; B[2, 1::2] = b
    d = np.array([[2 * n, 2 * n + 1] for n in tri]).ravel()
    K[np.ix_(d, d)] += A2 / 2 * B.T @ D @ B
u = np.zeros(8); u[5] = u[7] = 1e-4          # dy[TopNodes] = 1e-4, everything else constrained to 0
json.dump({"K": K.tolist(), "u": u.tolist(), "f_int": (K @ u).tolist(), "lodi_load_dy_TopNodes": float((K @ u)[[5, 7]].sum())},
          open(__file__.replace("make_expected.py", "expected.json"), "w"), indent=1)
