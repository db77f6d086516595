"""How many sorted-list insertions does a wave of the four-lanes-per-point kNN (csrc/vn_common.hpp, vn_knn_quad) EXECUTE per cloud,
and would a deeper pending queue or a seeding pass lower that?  (1.07 M VALU wave-instructions per cloud, most of
them insertion.)  A wave holds 16 points x 4 lanes; an insertion round is executed by the whole wave whenever any of its lanes
has an entry in the slot, so what counts is max-over-lanes, not the mean.  Model: normal clouds of 1024 points, k = 20, the
kernel's scan order (lane q takes candidates j0 + q + {0, 4, 8, 12}), its drain rule (any lane's queue could overflow on the next
group) and its skip of empty rounds.

    python tools/knn_drain_model.py

Result (round 4, 4 clouds per row): 216 executed insertions per wave with the product's 8 slots; 206-215 with 12-16 slots, MORE
with 24-64 (the threshold goes stale: 150-210 entries queued per point instead of 115, most of them no-ops when drained);
seeding the list with the first 32-64 candidates at full lane efficiency: 207-231.  Nothing here moves the count by more than
5 %: the insertion work is set by k (1 + ln(N / k)) = 99 accepted candidates per point and by the spread of that count over
the 64 lanes, not by the queue's depth.
"""
import numpy as np

N, K = 1024, 20


def sim(rng, slots, seed_n=0, trials=4):
    tot_exec = tot_offers = tot_drains = 0
    for _ in range(trials):
        pts = rng.standard_normal((N, 3)).astype(np.float32)
        sq = (pts ** 2).sum(1)
        wave_pts = rng.choice(N, 16, replace=False)
        sc = -(sq[None, :]) + 2 * pts[wave_pts] @ pts.T - sq[wave_pts][:, None]   # (16, N): the reference's score
        lists = [[] for _ in range(16)]
        thr = np.full(16, -np.inf)

        def ins(p, v):
            L = lists[p]
            if len(L) < K or v > L[-1]:
                L.append(v)
                L.sort(reverse=True)
                if len(L) > K:
                    L.pop()

        start, ex = 0, 0
        if seed_n:   # the first seed_n candidates inserted directly: one executed round per candidate, every lane busy
            for j in range(seed_n):
                for p in range(16):
                    ins(p, sc[p, j])
            ex += seed_n
            for p in range(16):
                thr[p] = lists[p][-1] if len(lists[p]) >= K else -np.inf
            start = seed_n
        cnt = np.zeros((16, 4), int)
        queue = [[[] for _ in range(4)] for _ in range(16)]
        nd = 0
        for j0 in list(range(start, N, 16)) + [None]:
            if j0 is not None:
                for q in range(4):
                    for m in range(4):
                        j = j0 + q + 4 * m
                        for p in range(16):
                            if sc[p, j] > thr[p]:
                                queue[p][q].append(sc[p, j])
                                cnt[p, q] += 1
                                tot_offers += 1
            if j0 is None or cnt.max() > slots - 4:
                for s in range(cnt.max()):
                    for q in range(4):
                        if (cnt[:, q] > s).any():     # the round runs for the whole wave
                            ex += 1
                            for p in range(16):
                                if cnt[p, q] > s:
                                    ins(p, queue[p][q][s])
                for p in range(16):
                    for q in range(4):
                        queue[p][q] = []
                    thr[p] = lists[p][-1] if len(lists[p]) >= K else -np.inf
                cnt[:] = 0
                nd += 1
        tot_exec += ex
        tot_drains += nd
    return tot_exec / trials, tot_offers / trials / 16, tot_drains / trials


def main():
    rng = np.random.default_rng(1)
    for slots, seed in [(8, 0), (12, 0), (16, 0), (24, 0), (32, 0), (8, 32), (8, 64), (16, 32), (16, 64), (24, 64), (32, 64), (64, 64)]:
        e, o, d = sim(rng, slots, seed)
        print(f"slots {slots:3d} seed {seed:3d}: executed insertions per wave {e:7.1f}  queued per point {o:6.1f}  drains {d:5.1f}")


if __name__ == "__main__":
    main()
