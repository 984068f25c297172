"""Coefficients of the kernel's cos(2 pi u) = (-1)^k cos(pi w), k = rint(2u), w = 2u - k in [-1/2, 1/2]:
cos(pi w) = 1 + z*R(z), z = w^2 in [0, 1/4], R fitted by Chebyshev interpolation in extended precision;
prints the double coefficients and the achieved error."""
import numpy as np

ld = np.longdouble
pi = ld("3.14159265358979323846264338327950288")


def q_exact(z):
    z = np.asarray(z, dtype=ld)
    out = np.zeros_like(z)
    zk = np.ones_like(z)
    fact = ld(2)
    for k in range(1, 40):
        out = out + (-1) ** k * pi ** (2 * k) / fact * zk
        zk = zk * z
        fact = fact * (2 * k + 1) * (2 * k + 2)
    return out


def fit(deg, n=400):
    theta = ld(np.pi) * (np.arange(n, dtype=ld) + ld(0.5)) / n
    theta = pi * (np.arange(n).astype(ld) + ld(0.5)) / n
    x = np.cos(theta)
    f = q_exact((x + 1) * ld(0.125))
    c = np.array([(2.0 / n) * np.sum(f * np.cos(j * theta)) for j in range(deg + 1)], dtype=ld)
    c[0] = c[0] / 2
    # Chebyshev -> monomials in x (Clenshaw on polynomials), then x = 8 z - 1
    T = [np.array([1], dtype=ld), np.array([0, 1], dtype=ld)]
    for j in range(2, deg + 1):
        t = np.zeros(j + 1, dtype=ld)
        t[1:] += 2 * T[-1]
        t[:j - 1] -= T[-2]
        T.append(t)
    px = np.zeros(deg + 1, dtype=ld)
    for j in range(deg + 1):
        px[:j + 1] += c[j] * T[j]
    pz = np.zeros(deg + 1, dtype=ld)
    base = np.array([1], dtype=ld)
    for a in px:
        pz[:len(base)] += a * base
        nb = np.zeros(len(base) + 1, dtype=ld)
        nb[:-1] -= base
        nb[1:] += 8 * base
        base = nb
    return pz


def kernel(u, coef):
    u = np.asarray(u, dtype=np.float64)
    t = 2.0 * u
    k = np.rint(t)
    w = t - k
    z = w * w
    q = np.full_like(z, coef[-1])
    for a in coef[-2::-1]:
        q = q * z + a
    c = z * q + 1.0
    return np.where(k == 1, -c, c)


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    u = np.concatenate([rng.random(2000000), 2.0 ** -rng.integers(1, 53, 5000) * rng.random(5000), [0, 0.25, 0.5, 0.75, 0.125]])
    want = np.cos(2 * pi * u.astype(ld))
    for deg in (6, 7, 8):
        coef = np.asarray(fit(deg), dtype=np.float64)
        err = float(np.max(np.abs(kernel(u, coef).astype(ld) - want)))
        print(deg, "max abs err %.3e" % err)
        if deg == 7:
            for a in coef:
                print("    %.20e," % a)
