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

if "--check" in sys.argv:
    rng = np.random.default_rng(0)
    z = rng.standard_normal(48) + 1j * rng.standard_normal(48)
    env = {"re": z.real.astype(np.float32), "im": z.imag.astype(np.float32)}
    for dst, expr in lines:
        env[dst] = eval(expr.replace("f", ""), {"__builtins__": {}}, dict(env, np=np)) if False else None
    # evaluate with float32 semantics
    vals = {}
    def get(tok):
        if tok.startswith("re["): return np.float32(env["re"][int(tok[3:-1])])
        if tok.startswith("im["): return np.float32(env["im"][int(tok[3:-1])])
        return vals[tok]
    for dst, expr in lines:
        toks = expr.split()
        if len(toks) == 1 and toks[0].startswith("-"):
            vals[dst] = -get(toks[0][1:])
        elif len(toks) == 3:
            a_, op, b_ = toks
            bv = np.float32(float(b_[:-1])) if b_.endswith("f") and b_[0] in "-0123456789" else get(b_)
            av = get(a_)
            vals[dst] = {"+": av + bv, "-": av - bv, "*": av * bv}[op]
        elif len(toks) == 5:
            a_, o1, b_, o2, c_ = toks
            if o1 == "*":      # a * const + c
                vals[dst] = np.float32(get(a_) * np.float32(float(b_[:-1]))) + get(c_)
            else:              # b - a * const
                vals[dst] = get(a_) - np.float32(get(c_[:0] or b_) * 1) if False else get(a_) - np.float32(get(b_) * np.float32(float(c_[:-1])))
        else:
            raise SystemExit("cannot parse " + expr)
    got = np.array([complex(get(o[0]), get(o[1])) for o in out])
    want = np.fft.fft(z.astype(np.complex64).astype(np.complex128))
    print("ops", len(lines), "max err vs numpy.fft:", np.abs(got - want).max(), "scale", np.abs(want).max())
    sys.exit(0)

with open("equiadapt_amd/csrc/fft48.inc", "w") as f:
    f.write("// GENERATED by tools/gen_fft48.py -- do not edit.  48-point complex DFT, forward sign, natural order in and out.\n")
    f.write("// %d floating-point operations, all indices static.  Inverse: call with (im, re) in and (oim, ore) out.\n" % len(lines))
    f.write("__device__ __forceinline__ void fft48(const float (&re)[48], const float (&im)[48], float (&ore)[48], float (&oim)[48]) {\n")
    for dst, expr in lines:
        f.write(f"  const float {dst} = {expr};\n")
    for k, (r, i) in enumerate(out):
        f.write(f"  ore[{k}] = {r}; oim[{k}] = {i};\n")
    f.write("}\n")
print("wrote fft48.inc with", len(lines), "operations")
