import numpy as np
from scipy.special import erf, ndtr
from scipy.optimize import least_squares
x = np.linspace(-9, 9, 36001)
gelu = x * ndtr(x)
def model(c, x):
    x2 = x * x
    u = x * (c[0] + x2 * (c[1] + x2 * (c[2] + (x2 * c[3] if len(c) > 3 else 0))))
    return x / (1 + np.exp(-u))
for n in (2, 3, 4):
    c0 = [1.5957691, 0.0713548, 0.0, 0.0][:n] if n > 2 else [1.5957691, 0.0713548]
    def m2(c, x):
        x2 = x * x
        u = c[-1]
        for k in range(len(c) - 2, -1, -1):
            u = u * x2 + c[k]
        return x / (1 + np.exp(-np.clip(u * x, -80, 80)))
    best = None
    # minimax via iteratively reweighted least squares
    w = np.ones_like(x)
    c = np.array(c0, float)
    for it in range(60):
        r = least_squares(lambda c: w * (m2(c, x) - gelu), c, xtol=1e-15, ftol=1e-15)
        c = r.x
        e = np.abs(m2(c, x) - gelu)
        w = w * (1 + 3 * e / e.max())
        w /= w.mean()
    e = np.abs(m2(c, x) - gelu)
    print(n, "coeffs", [f"{v:.10g}" for v in c], "max abs err", e.max(), "at", x[e.argmax()])
    # fp32 emulation
    xf = x.astype(np.float32); cf = c.astype(np.float32)
    x2 = xf * xf
    u = cf[-1]
    for k in range(len(cf) - 2, -1, -1):
        u = (u * x2 + cf[k]).astype(np.float32)
    u = (u * xf).astype(np.float32)
    g32 = xf / (np.float32(1) + np.exp2((-u * np.float32(1.4426950408889634)).astype(np.float32)))
    print("   fp32 max abs err", np.abs(g32.astype(np.float64) - gelu).max())
# reference: tanh approx
t = 0.5 * x * (1 + np.tanh(np.sqrt(2 / np.pi) * (x + 0.044715 * x ** 3)))
print("tanh-approx max abs err", np.abs(t - gelu).max())
# current A&S
z = np.abs(x) * 0.70710678118654752
tt = 1 / (1 + 0.3275911 * z)
poly = tt * (0.254829592 + tt * (-0.284496736 + tt * (1.421413741 + tt * (-1.453152027 + tt * 1.061405429))))
e_ = 1 - poly * np.exp(-z * z)
print("A&S 7.1.26 max abs err", np.abs(0.5 * x * (1 + np.copysign(e_, x)) - gelu).max())
