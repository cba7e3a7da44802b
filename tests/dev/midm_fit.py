"""dev (round 5): least-squares refit of the prefill planner's launch model (gptqhip_tiled.hip:tiled_cost_us) from the
`P K N M bm s us` lines of tests/dev/midm_heights.py (several files = several runs: points are averaged), one coefficient row per tile
height + one shared idle-CU term, and a replay of the planner's search with the new coefficients against the best measured point
of every (shape, M).  Prints the C++ table."""
import collections
import sys

import numpy as np

raw = collections.defaultdict(list)
auto = collections.defaultdict(list)
for fn in sys.argv[1:]:
    for ln in open(fn):
        if not ln.startswith("P "):
            continue
        f = ln.split("#")[0].split()
        K, N, M, bm, s = map(int, f[1:6])
        (auto[(K, N, M)] if bm == 0 else raw[(K, N, M, bm, s)]).append(float(f[6]))
meas = {k: float(np.mean(v)) for k, v in raw.items()}
HS = sorted({k[3] for k in meas})
RED_MIN, RED_A, RED_BW, IDLE_AT = 4.6, 1.5, 6.2, 0.75


def feats(K, N, M, bm, s):
    # bm >= 1000: the 128-column-block form with tile height bm - 1000 (tests/dev/midm_heights.py with BN=128)
    bn, bm = (128, bm - 1000) if bm >= 1000 else (256, bm)
    chunks = -(-K // 128)
    cps = -(-chunks // s)
    s_eff = -(-chunks // cps)
    tiles = -(-N // bn) * -(-M // bm)
    B = tiles * s_eff
    if s_eff == 1:
        full, rem = divmod(tiles, 256)
        r = full + ((0.75 + 0.25 * rem / 256) if rem else 0.0)
        f = 1.0 if full >= 1 else rem / 256
    else:
        r = 1.0 if B <= 256 else B / 256
        f = min(B, 256) / 256
    red = max(RED_MIN, RED_A + s_eff * M * N * 4 / 1e6 / RED_BW) if s_eff > 1 else 0.0
    return [1.0, f, r * cps, r * cps * f], max(0.0, IDLE_AT - f), red


A, y, w = [], [], []
for (K, N, M, bm, s), us in meas.items():
    x, idle, red = feats(K, N, M, bm, s)
    row = [0.0] * (4 * len(HS)) + [idle]
    i = HS.index(bm)
    row[4 * i:4 * i + 4] = x
    A.append(row)
    y.append(us - red)
    w.append(1.0 / us)
A, y, w = np.array(A), np.array(y), np.array(w)
c, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
err = np.abs(A @ c - y) * w
print(f"{len(y)} points, relative fit error mean {100 * err.mean():.1f} %, p90 {100 * np.percentile(err, 90):.1f} %")
print("    static const double kCoef[%d][4] = {" % len(HS) + ", ".join("{%.3f, %.3f, %.3f, %.3f}" % tuple(c[4 * i:4 * i + 4]) for i in range(len(HS))) + "};")
print(f"    heights {HS}; kIdle = {c[-1]:.3f}")


def predict(K, N, M, bm, s):
    x, idle, red = feats(K, N, M, bm, s)
    i = HS.index(bm)
    return float(np.dot(c[4 * i:4 * i + 4], x) + c[-1] * idle + red)


best = {}
for (K, N, M, bm, s), us in meas.items():
    if (K, N, M) not in best or us < best[(K, N, M)][0]:
        best[(K, N, M)] = (us, bm, s)
loss, loss_auto = [], []
for key in sorted(best):
    cands = [(predict(*k), k[3], k[4]) for k in meas if k[:3] == key]
    pred, bm, s = min(cands)
    got = meas[key + (bm, s)]
    loss.append(got / best[key][0] - 1)
    a = float(np.mean(auto[key])) if key in auto else float("nan")
    loss_auto.append(a / best[key][0] - 1)
    flag = "  <--" if loss[-1] > 0.04 else ""
    print(f"K={key[0]:5d} N={key[1]:5d} M={key[2]:4d}: model picks bm={bm:3d} s={s:2d} pred {pred:6.1f} meas {got:6.1f} | best {best[key][0]:6.1f} (bm={best[key][1]}, s={best[key][2]}) | shipped auto {a:6.1f}{flag}")
print(f"model-vs-best: mean loss {100 * np.mean(loss):.1f} %, max {100 * np.max(loss):.1f} %; shipped-auto-vs-best: mean {100 * np.nanmean(loss_auto):.1f} %, max {100 * np.nanmax(loss_auto):.1f} %")
# residual structure: by chunks-per-block (power-of-two strides between the K slices of concurrent blocks?)
res = collections.defaultdict(list)
for (K, N, M, bm, s), us in meas.items():
    chunks = -(-K // 128)
    cps = -(-chunks // s)
    if s > 1:
        res[cps].append((us - predict(K, N, M, bm, s)) / us)
print("relative residual by chunks per block:", {k: (round(100 * float(np.mean(v)), 1), len(v)) for k, v in sorted(res.items())})
