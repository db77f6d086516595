"""CPU model of the distributed sorted-list insertion of the four-lanes-per-point kNN (csrc/vn_common.hpp, vn_knn_quad):
lane q of a quad holds ranks [SEG q, SEG q + SEG); a candidate (v, j) seen by all four lanes is inserted by every lane at once,
each using the OLD last entry of its predecessor.  Checked against a stable descending sort (earlier insertion wins ties).

    python tools/knn_quad_model.py
"""
import numpy as np


def insert(S, I, v, j, SEG):
    pv = [np.inf] + [S[q - 1][SEG - 1] for q in range(1, 4)]
    pi = [0] + [I[q - 1][SEG - 1] for q in range(1, 4)]
    for q in range(4):
        take = v > pv[q]                    # the candidate lands before this segment: everything moves down, pv comes in
        cv, ci = (pv[q], pi[q]) if take else (v, j)
        g = [take or (cv > S[q][t]) for t in range(SEG)]   # monotone: the segment is sorted
        for t in range(SEG - 1, 0, -1):     # every new entry is a select of OLD values
            if g[t]:
                S[q][t], I[q][t] = (S[q][t - 1], I[q][t - 1]) if g[t - 1] else (cv, ci)
        if g[0]:
            S[q][0], I[q][0] = cv, ci


def main():
    rng = np.random.default_rng(0)
    for SEG in (5, 8):
        K = 4 * SEG
        for trial in range(300):
            n = int(rng.integers(1, 300))
            vals = rng.standard_normal(n).astype(np.float32)
            if trial % 3 == 0:
                vals = np.round(vals * 4) / 4   # many exact ties
            S = [[-np.inf] * SEG for _ in range(4)]
            I = [[0] * SEG for _ in range(4)]
            for j, v in enumerate(vals):
                insert(S, I, v, j, SEG)
            got_i = [x for s in I for x in s]
            order = np.argsort(-vals, kind="stable")[:K]
            m = min(n, K)
            assert got_i[:m] == list(order[:m]), (SEG, trial)
    print("distributed insertion == stable top-k for SEG in (5, 8), 600 random streams incl. ties")


if __name__ == "__main__":
    main()
