"""Generates equiadapt_amd/csrc/fft48.inc: a straight-line 48-point complex DFT (forward, e^{-2 pi i nk/48}) for registers.

48 = 3 x 16 = 3 x 4 x 4 (Cooley-Tukey, decimation in time): 16 radix-3 butterflies, twiddles W48^{n2 k1}, three 16-point
transforms of two radix-4 passes each.  All twiddles are literals, every index is static.  `python tools/gen_fft48.py` rewrites
the file; `python tools/gen_fft48.py --check` runs the same operation list in numpy against numpy.fft.fft.
The inverse transform is the forward one with real and imaginary parts swapped on the way in and out (unscaled)."""
import math, sys
import numpy as np

lines = []   # (dst, op, a, b, const)
def emit(dst, expr):
    lines.append((dst, expr))

cnt = [0]
def tmp():
    cnt[0] += 1
    return f"t{cnt[0]}"

def add(a, b):
    d = tmp(); emit(d, f"{a} + {b}"); return d
def sub(a, b):
    d = tmp(); emit(d, f"{a} - {b}"); return d
def mulc(a, c):
    d = tmp(); emit(d, f"{a} * {lit(c)}"); return d
def fma(a, c, b):   # a*c + b
    d = tmp(); emit(d, f"{a} * {lit(c)} + {b}"); return d
def fms(a, c, b):   # b - a*c
    d = tmp(); emit(d, f"{b} - {a} * {lit(c)}"); return d
def lit(c):
    return repr(float(np.float32(c))) + "f"

def cadd(x, y): return (add(x[0], y[0]), add(x[1], y[1]))
def csub(x, y): return (sub(x[0], y[0]), sub(x[1], y[1]))
def cmul_const(x, m, n):
    """x * exp(-2 pi i m / n)"""
    m %= n
    if m == 0: return x
    c, s = math.cos(2 * math.pi * m / n), -math.sin(2 * math.pi * m / n)   # W = c + i s
    if 4 * m == n: return (x[1], neg(x[0]))            # -i: (a+ib)(-i) = b - i a
    if 2 * m == n: return (neg(x[0]), neg(x[1]))
    if 4 * m == 3 * n: return (neg(x[1]), x[0])        # +i
    re = fms(x[1], s, mulc(x[0], c))                    # a c - b s
    im = fma(x[0], s, mulc(x[1], c))                    # a s + b c
    return (re, im)
def neg(a):
    d = tmp(); emit(d, f"-{a}"); return d

def radix3(x0, x1, x2):
    t1 = cadd(x1, x2)
    a0 = cadd(x0, t1)
    t2 = (fms(t1[0], 0.5, x0[0]), fms(t1[1], 0.5, x0[1]))
    d = csub(x1, x2)
    t3 = (mulc(d[0], math.sqrt(3) / 2), mulc(d[1], math.sqrt(3) / 2))
    a1 = (add(t2[0], t3[1]), sub(t2[1], t3[0]))        # t2 - i t3
    a2 = (sub(t2[0], t3[1]), add(t2[1], t3[0]))        # t2 + i t3
    return a0, a1, a2

def radix4(y0, y1, y2, y3):
    e, f, g, h = cadd(y0, y2), csub(y0, y2), cadd(y1, y3), csub(y1, y3)
    c0, c2 = cadd(e, g), csub(e, g)
    c1 = (add(f[0], h[1]), sub(f[1], h[0]))            # f - i h
    c3 = (sub(f[0], h[1]), add(f[1], h[0]))            # f + i h
    return c0, c1, c2, c3

def fft16(y):
    c = [[None] * 4 for _ in range(4)]                 # c[q1][m2]
    for m2 in range(4):
        r = radix4(y[m2], y[4 + m2], y[8 + m2], y[12 + m2])
        for q1 in range(4):
            c[q1][m2] = cmul_const(r[q1], m2 * q1, 16)
    X = [None] * 16
    for q1 in range(4):
        r = radix4(*c[q1])
        for q2 in range(4):
            X[q1 + 4 * q2] = r[q2]
    return X

def build():
    x = [(f"re[{n}]", f"im[{n}]") for n in range(48)]
    a = [[None] * 16 for _ in range(3)]
    for n2 in range(16):
        r = radix3(x[n2], x[16 + n2], x[32 + n2])
        for k1 in range(3):
            a[k1][n2] = cmul_const(r[k1], n2 * k1, 48)
    out = [None] * 48
    for k1 in range(3):
        X = fft16(a[k1])
        for k2 in range(16):
            out[k1 + 3 * k2] = X[k2]
    return out

out = build()
lines_fft = lines


def fft8(y):
    c = [[None] * 2 for _ in range(4)]                 # c[q1][m2]
    for m2 in range(2):
        r = radix4(y[m2], y[2 + m2], y[4 + m2], y[6 + m2])
        for q1 in range(4):
            c[q1][m2] = cmul_const(r[q1], m2 * q1, 8)
    X = [None] * 8
    for q1 in range(4):
        X[q1], X[q1 + 4] = cadd(c[q1][0], c[q1][1]), csub(c[q1][0], c[q1][1])
    return X


def fft24(x):
    a = [[None] * 8 for _ in range(3)]
    for n2 in range(8):
        r = radix3(x[n2], x[8 + n2], x[16 + n2])
        for k1 in range(3):
            a[k1][n2] = cmul_const(r[k1], n2 * k1, 24)
    o = [None] * 24
    for k1 in range(3):
        X = fft8(a[k1])
        for k2 in range(8):
            o[k1 + 3 * k2] = X[k2]
    return o


def build_c2r():
    """Real output x[0..47] = sum over all 48 frequencies of a Hermitian spectrum given by its half X[0..24] (unscaled inverse,
    e^{+2 pi i kn/48}; im[0] and im[24] are ignored as in numpy.fft.irfft).  Half-length algorithm: with
    E_k = X_k + conj(X_{24-k}), O_k = (X_k - conj(X_{24-k})) e^{+2 pi i k/48}, Z_k = E_k + i O_k (k = 0..23; Z_{24-k} follows
    from the same E_k, O_k), the 24-point inverse transform of Z is x[2n] + i x[2n+1]."""
    X = [(f"re[{k}]", f"im[{k}]") for k in range(25)]
    Z = [None] * 24
    Z[0] = (add(X[0][0], X[24][0]), sub(X[0][0], X[24][0]))
    for k in range(1, 12):
        A, B = X[k], X[24 - k]
        E = (add(A[0], B[0]), sub(A[1], B[1]))
        D = (sub(A[0], B[0]), add(A[1], B[1]))
        O = cmul_const(D, 48 - k, 48)
        Z[k] = (sub(E[0], O[1]), add(E[1], O[0]))
        Z[24 - k] = (add(E[0], O[1]), sub(O[0], E[1]))
    Z[12] = (add(X[12][0], X[12][0]), neg(add(X[12][1], X[12][1])))
    o = fft24([(zi, zr) for (zr, zi) in Z])            # inverse = forward with re / im swapped on the way in and out
    x = [None] * 48
    for n in range(24):
        x[2 * n], x[2 * n + 1] = o[n][1], o[n][0]
    return x


lines = []
out_c2r = build_c2r()
lines_c2r = lines


def cscale(x, c):
    return (mulc(x[0], c), mulc(x[1], c))


def build_r2c():
    """Half spectrum X[0..24] (forward sign, unscaled) of the real input re[0..47]: z_n = x[2n] + i x[2n+1], Z = 24-point
    transform of z, X_k = (Z_k + conj Z_{24-k}) / 2 - i e^{-2 pi i k/48} (Z_k - conj Z_{24-k}) / 2, and X_{24-k} from the same
    two terms; X_0 = Re Z_0 + Im Z_0, X_24 = Re Z_0 - Im Z_0 (both real), X_12 = conj Z_12."""
    z = [(f"re[{2 * n}]", f"re[{2 * n + 1}]") for n in range(24)]
    Z = fft24(z)
    X = [None] * 25
    X[0] = (add(Z[0][0], Z[0][1]), "0.0f")
    X[24] = (sub(Z[0][0], Z[0][1]), "0.0f")
    X[12] = (Z[12][0], neg(Z[12][1]))
    for k in range(1, 12):
        A, B = Z[k], Z[24 - k]
        S = (add(A[0], B[0]), sub(A[1], B[1]))            # A + conj B
        T = (sub(A[0], B[0]), add(A[1], B[1]))            # A - conj B
        # U = -i w^k T / 2,  w = e^{-2 pi i/48}:  -i w^k = e^{-2 pi i (k + 12)/48}
        c, sn = 0.5 * math.cos(2 * math.pi * (k + 12) / 48), -0.5 * math.sin(2 * math.pi * (k + 12) / 48)
        U = (fms(T[1], sn, mulc(T[0], c)), fma(T[0], sn, mulc(T[1], c)))
        Hh = cscale(S, 0.5)
        X[k] = cadd(Hh, U)
        X[24 - k] = (sub(Hh[0], U[0]), sub(U[1], Hh[1]))  # conj(S/2 - U)
    return X


lines = []
out_r2c = build_r2c()
lines_r2c = lines


def build_half_inverse(parity):
    """Rows y = 2 r + parity (r = 0..23) of the unscaled 48-point INVERSE transform x[y] = sum_k X_k e^{+2 pi i k y/48} by one
    decimation-in-frequency step: a 24-point inverse transform of X_k + X_{k+24} (even rows) or of
    (X_k - X_{k+24}) e^{+2 pi i k/48} (odd rows)."""
    X = [(f"re[{k}]", f"im[{k}]") for k in range(48)]
    Z = []
    for k in range(24):
        if parity == 0:
            Z.append(cadd(X[k], X[k + 24]))
        else:
            Z.append(cmul_const(csub(X[k], X[k + 24]), 48 - k, 48))   # e^{+2 pi i k/48} = e^{-2 pi i (48-k)/48}
    o = fft24([(zi, zr) for (zr, zi) in Z])            # inverse = forward with re / im swapped on the way in and out
    return [(o[r][1], o[r][0]) for r in range(24)]


lines = []
out_even = build_half_inverse(0)
lines_even = lines
lines = []
out_odd = build_half_inverse(1)
lines_odd = lines


def evaluate(z, ops=None, outs=None):
    """Run an operation list in numpy with float32 rounding after every operation (what the device does without FMA
    contraction) on the complex vector z; default: the 48-point transform."""
    ops = lines_fft if ops is None else ops
    vals = {}
    re, im = z.real.astype(np.float32), z.imag.astype(np.float32)

    def get(tok):
        if tok.startswith("re["): return re[int(tok[3:-1])]
        if tok.startswith("im["): return im[int(tok[3:-1])]
        if tok[0] in "-0123456789" and tok.endswith("f"): return np.float32(float(tok[:-1]))
        return vals[tok]

    for dst, expr in ops:
        t = expr.split()
        if len(t) == 1:                                   # -a
            vals[dst] = -get(t[0][1:])
        elif len(t) == 3:                                 # a op b
            a_, b_ = get(t[0]), get(t[2])
            vals[dst] = {"+": a_ + b_, "-": a_ - b_, "*": np.float32(a_ * b_)}[t[1]]
        elif t[1] == "*":                                 # a * c + b
            vals[dst] = np.float32(get(t[0]) * get(t[2])) + get(t[4])
        else:                                             # b - a * c
            vals[dst] = get(t[0]) - np.float32(get(t[2]) * get(t[4]))
    if outs is not None:
        return np.array([get(r) for r in outs])
    return np.array([complex(get(r), get(i)) for r, i in out])


def render():
    o = ["// GENERATED by tools/gen_fft48.py -- do not edit.  48-point complex DFT, forward sign, natural order in and out.",
         "// %d floating-point operations, all indices static.  Inverse: call with (im, re) in and (oim, ore) out." % len(lines_fft),
         "// T: float, or a 2-vector of floats (two independent transforms per lane on the packed fp32 instructions).",
         "template <typename T>",
         "__device__ __forceinline__ void fft48(const T (&re)[48], const T (&im)[48], T (&ore)[48], T (&oim)[48]) {"]
    o += [f"  const T {dst} = {expr};" for dst, expr in lines_fft]
    o += [f"  ore[{k}] = {r}; oim[{k}] = {i};" for k, (r, i) in enumerate(out)]
    o += ["}", "",
          "// Real 48-point output of a Hermitian spectrum from its half re/im[0..24] (unscaled inverse; im[0], im[24] ignored):",
          "// one 24-point complex transform of Z_k = E_k + i O_k (see tools/gen_fft48.py build_c2r).  %d operations." % len(lines_c2r),
          "template <typename T>",
          "__device__ __forceinline__ void ifft48_c2r(const T (&re)[25], const T (&im)[25], T (&x)[48]) {"]
    o += [f"  const T {dst} = {expr};" for dst, expr in lines_c2r]
    o += [f"  x[{n}] = {v};" for n, v in enumerate(out_c2r)]
    o += ["}", "",
          "// Half spectrum ore/oim[0..24] (forward sign, unscaled) of the real input re[0..47]: one 24-point complex transform of",
          "// x[2n] + i x[2n+1] and a recombination pass (see tools/gen_fft48.py build_r2c).  %d operations." % len(lines_r2c),
          "template <typename T>",
          "__device__ __forceinline__ void fft48_r2c(const T (&re)[48], T (&ore)[25], T (&oim)[25]) {"]
    o += [f"  const T {dst} = {expr};" for dst, expr in lines_r2c]
    o += [f"  ore[{k}] = {r}; oim[{k}] = {'T(0.0f)' if i == '0.0f' else i};" for k, (r, i) in enumerate(out_r2c)]
    o += ["}"]
    # the half-length inverse transforms were used by an experiment (a tile's rows split by parity over two co-resident blocks:
    # slower, HISTORY.md section 6); `--half` emits them again
    halves = (("ifft48_even", lines_even, out_even, "even rows y = 2 r"), ("ifft48_odd", lines_odd, out_odd, "odd rows y = 2 r + 1"))
    for name, ls, outs, what in (halves if "--half" in sys.argv else ()):
        o += ["",
              "// The %s (r = 0..23) of the unscaled 48-point inverse transform: one decimation-in-frequency step and a 24-point" % what,
              "// inverse transform (see tools/gen_fft48.py build_half_inverse).  %d operations." % len(ls),
              "template <typename T>",
              "__device__ __forceinline__ void %s(const T (&re)[48], const T (&im)[48], T (&ore)[24], T (&oim)[24]) {" % name]
        o += [f"  const T {dst} = {expr};" for dst, expr in ls]
        o += [f"  ore[{k}] = {r}; oim[{k}] = {i};" for k, (r, i) in enumerate(outs)]
        o += ["}"]
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    if "--check" in sys.argv:
        rng = np.random.default_rng(0)
        z = rng.standard_normal(48) + 1j * rng.standard_normal(48)
        got = evaluate(z)
        want = np.fft.fft(z.astype(np.complex64).astype(np.complex128))
        print("ops", len(lines_fft), "max err vs numpy.fft:", np.abs(got - want).max(), "scale", np.abs(want).max())
        ok = np.abs(got - want).max() < 1e-5 * np.abs(want).max()
        h = (rng.standard_normal(25) + 1j * rng.standard_normal(25)).astype(np.complex64).astype(np.complex128)
        gotr = evaluate(h, lines_c2r, out_c2r)
        wantr = np.fft.irfft(h, 48) * 48
        print("c2r ops", len(lines_c2r), "max err vs numpy.fft.irfft:", np.abs(gotr - wantr).max(), "scale", np.abs(wantr).max())
        ok = ok and np.abs(gotr - wantr).max() < 1e-5 * np.abs(wantr).max()
        xr = rng.standard_normal(48).astype(np.float32)
        goth = evaluate(xr.astype(np.complex128), lines_r2c, [t for pair in out_r2c for t in pair]).reshape(25, 2)
        wanth = np.fft.rfft(xr.astype(np.float64))
        errh = np.abs(goth[:, 0] + 1j * goth[:, 1] - wanth).max()
        print("r2c ops", len(lines_r2c), "max err vs numpy.fft.rfft:", errh, "scale", np.abs(wanth).max())
        ok = ok and errh < 1e-5 * np.abs(wanth).max()
        zz = (rng.standard_normal(48) + 1j * rng.standard_normal(48)).astype(np.complex64).astype(np.complex128)
        full = np.fft.ifft(zz) * 48
        for par, ls, outs in ((0, lines_even, out_even), (1, lines_odd, out_odd)):
            g = evaluate(zz, ls, [t for pair in outs for t in pair]).reshape(24, 2)
            e = np.abs(g[:, 0] + 1j * g[:, 1] - full[par::2]).max()
            print("half inverse parity", par, "ops", len(ls), "max err vs numpy.fft.ifft:", e, "scale", np.abs(full).max())
            ok = ok and e < 1e-5 * np.abs(full).max()
        sys.exit(0 if ok else 1)
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "equiadapt_amd", "csrc", "fft48.inc")
    with open(path, "w") as f:
        f.write(render())
    print("wrote", path, "with", len(lines_fft), "+", len(lines_c2r), "+", len(lines_r2c), "operations")
