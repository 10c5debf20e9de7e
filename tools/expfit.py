import mpmath as mp, numpy as np
mp.mp.dps = 50
a = mp.log(2)/2 * mp.mpf('1.0001')
for deg in (10, 11, 12):
    # fit (exp(r)-1-r)/r^2 ? simpler: fit exp directly
    c, err = mp.chebyfit(mp.exp, [-a, a], deg+1, error=True)
    c = c[::-1]  # ascending
    cd = [float(x) for x in c]
    print(deg, 'fit err', mp.nstr(err, 5))
    # evaluate double Horner vs exact on random pts
    rs = np.random.default_rng(0).uniform(-float(a), float(a), 20000)
    p = np.zeros_like(rs) + cd[-1]
    for k in range(deg-1, -1, -1):
        p = p*rs + cd[k]   # not fma, but close
    ex = np.array([float(mp.exp(mp.mpf(r))) for r in rs])
    print('   max rel', np.max(np.abs(p-ex)/ex)/2.22e-16, 'ulp')
    print('   coeffs', [x.hex() for x in cd])
    print('   ', cd)


def sin_fit():
    """Odd minimax polynomial for sin on [-pi/2, pi/2] (rff_cos in csrc/rff.hip): sin(r) = r * P(r^2)."""
    a = (mp.pi / 2) ** 2 * mp.mpf('1.0002')

    def g(z):
        if z == 0:
            return mp.mpf(1)
        r = mp.sqrt(z)
        return mp.sin(r) / r
    c, err = mp.chebyfit(g, [0, a], 10, error=True)
    print('sin: fit err', mp.nstr(err, 5), [float(x).hex() for x in c[::-1]])


if __name__ == "__main__":
    sin_fit()
