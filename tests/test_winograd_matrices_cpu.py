"""The Winograd matrices the HIP kernels hard-code (conv_igemm.hip F(2x2,3x3), conv_wino4.hip F(4x4,3x3), conv_wino1d.hip F(2,5)), checked
in exact rational arithmetic against the plain correlation they replace -- host-side, no GPU: a typo in a coefficient would otherwise
only show up as a 1e-3 parity failure on the box."""
from fractions import Fraction as F
import random

# F(2,3): conv_wino_kernel / wino_weight_kernel (row 2 of B^T and of G are negated together in the kernel: exact, not modelled here)
BT23 = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
G23 = [[1, 0, 0], [F(1, 2), F(1, 2), F(1, 2)], [F(1, 2), F(-1, 2), F(1, 2)], [0, 0, 1]]
AT23 = [[1, 1, 1, 0], [0, 1, -1, -1]]
# six points 0, +-1, +-2, inf: B^T shared by F(4,3) (conv_wino4_kernel) and F(2,5) (conv_wino1d_kernel)
BT6 = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
PTS = [0, 1, -1, 2, -2]
SC = [F(1, 4), F(-1, 6), F(-1, 6), F(1, 24), F(1, 24)]


def g_matrix(r):
    return [[SC[i] * F(PTS[i]) ** k for k in range(r)] for i in range(5)] + [[0] * (r - 1) + [1]]


def at_matrix(m):
    return [[(F(PTS[i]) ** j if i < 5 else (1 if j == m - 1 else 0)) for i in range(6)] for j in range(m)]


def check_1d(AT, G, BT, m, r, trials=25):
    n = m + r - 1
    rnd = random.Random(m * 10 + r)
    for _ in range(trials):
        d = [F(rnd.randint(-9, 9)) for _ in range(n)]
        g = [F(rnd.randint(-9, 9)) for _ in range(r)]
        U = [sum(F(G[i][k]) * g[k] for k in range(r)) for i in range(n)]
        V = [sum(F(BT[i][j]) * d[j] for j in range(n)) for i in range(n)]
        y = [sum(F(AT[j][i]) * U[i] * V[i] for i in range(n)) for j in range(m)]
        assert y == [sum(d[j + k] * g[k] for k in range(r)) for j in range(m)]


def test_f2_3_is_exact():
    check_1d(AT23, G23, BT23, 2, 3)


def test_f4_3_is_exact():
    G = g_matrix(3)
    assert G[3] == [F(1, 24), F(1, 12), F(1, 6)] and G[1] == [F(-1, 6)] * 3          # Lavin's G
    AT = at_matrix(4)
    assert AT[3] == [0, 1, -1, 8, -8, 1]
    check_1d(AT, G, BT6, 4, 3)


def test_f2_5_is_exact():
    G = g_matrix(5)
    assert G[3] == [F(1, 24), F(1, 12), F(1, 6), F(1, 3), F(2, 3)]
    AT = at_matrix(2)
    assert AT == [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1]]                          # what conv_wino1d_kernel's tail applies
    check_1d(AT, G, BT6, 2, 5)


def test_f2_5_input_transform_factorisation():
    """conv_wino1d_kernel builds rows 0..2 / 3..5 of B^T d from shared sub-expressions; the same algebra in rationals."""
    rnd = random.Random(5)
    for _ in range(20):
        d = [F(rnd.randint(-9, 9)) for _ in range(6)]
        t, s = d[4] - 4 * d[2], d[3] - 4 * d[1]
        lo = [4 * d[0] + (t - d[2]), t + s, t - s]
        t2, s2 = d[4] - d[2], d[3] - d[1]
        hi = [2 * s2 + t2, -2 * s2 + t2, 4 * d[1] + (-5 * d[3] + d[5])]
        assert lo + hi == [sum(F(BT6[i][j]) * d[j] for j in range(6)) for i in range(6)]
