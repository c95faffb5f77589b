"""Development aid: what would a flow step last under other tail schedules?  Input: per-chain attempted steps of 8 flow steps
(tools/dump_ns.py).  Per-attempt costs in cycles from the stamps build (tools/flow_cycles.py)."""
import sys, os, heapq
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = np.load(os.path.join(ROOT, "tools", "data", "ns_keys.npz"))["ns"]
GHZ = 2.387
OTH = 12e3      # controller + stage inputs per attempt
C_FULL = 6 * 28.1e3 + 76.5e3 + OTH + 6e3
C_CMP8 = 6 * 26.9e3 + 52.6e3 + OTH
C_CMP3 = 6 * 26.9e3 + 17.6e3 + OTH
C_TAIL = 6 * 10.56e3 + 17.6e3 + 4e3
INIT = 4        # attempts a row spends in its two initial-step phases (2 per solve)

def tile_now(n):           # n: attempts of the 16 rows -> cycles of the tile as built today
    n = np.sort(n + INIT)[::-1]
    # attempts with > 8 live rows: n[8]; 4..8 live: n[3] - n[8]; 3 live: n[2] - n[3]; <= 2: n[0] - n[2]
    return n[8] * C_FULL + (n[3] - n[8]) * C_CMP8 + (n[2] - n[3]) * C_CMP3 + (n[0] - n[2]) * C_TAIL

def sim_fanout(nk, thr=8, helpers_stay=True, c_pair=C_TAIL, own_pairs=1):
    """tiles run full mode until <= thr rows live, then post their rows as pairs (sorted by remaining work? no: by row order);
    every workgroup that is free runs pairs.  A pair lasts max(rem_a, rem_b) * c_pair."""
    T = nk.reshape(-1, 16) + INIT
    W = T.shape[0]
    free = []          # (time a WG becomes free)
    items = []         # (post time, duration)
    for t in T:
        s = np.sort(t)[::-1]
        t0 = s[thr] * C_FULL
        rem = s[:thr] - s[thr]            # remaining attempts of the live rows, descending
        # pair them: longest with second longest etc. (rows of similar length share a pair)
        pairs = [rem[i] for i in range(0, thr, 2)]
        for p in pairs: items.append((t0, p * c_pair))
        free.append(t0)
    # greedy list scheduling: items in order of post time, longest first among equal; each WG free from its t0
    items.sort(key=lambda x: (x[0], -x[1]))
    h = list(free); heapq.heapify(h)
    end = 0.0
    # process in post-time order; a WG can take an item when free: start = max(free_time, post)
    for post, dur in items:
        f = heapq.heappop(h)
        st = max(f, post)
        heapq.heappush(h, st + dur)
        end = max(end, st + dur)
    return end

for j in range(ns.shape[0]):
    nk = ns[j].astype(float)
    now = max(tile_now(t) for t in nk.reshape(-1, 16))
    mean = np.mean([tile_now(t) for t in nk.reshape(-1, 16)])
    f8 = sim_fanout(nk, 8); f4 = sim_fanout(nk, 4); f12 = sim_fanout(nk, 12); f6 = sim_fanout(nk, 6)
    lb = (nk + INIT).max() * C_TAIL
    print(f"key {j}: now {now/GHZ/1e6:6.2f} ms (mean WG {mean/GHZ/1e6:6.2f}) | fan-out thr 4: {f4/GHZ/1e6:6.2f}  6: {f6/GHZ/1e6:6.2f}  8: {f8/GHZ/1e6:6.2f}  12: {f12/GHZ/1e6:6.2f} | slowest chain alone in tail mode {lb/GHZ/1e6:6.2f}")

print("\n-- time batch compacted by live rows in full mode: <= 12 rows -> 4 M tiles, <= 9 rows -> 3 M tiles")
for j in range(ns.shape[0]):
    nk = ns[j].astype(float)
    def tile2(n, c4=62e3 + 3e3, c3=52.6e3 + 3e3):
        n = np.sort(n + INIT)[::-1]
        full16 = n[12]; full12 = n[9] - n[12]; full9 = n[8] - n[9]
        return (full16 * C_FULL + full12 * (C_FULL - 76.5e3 + c4) + full9 * (C_FULL - 76.5e3 + c3)
                + (n[3] - n[8]) * C_CMP8 + (n[2] - n[3]) * C_CMP3 + (n[0] - n[2]) * C_TAIL)
    T = nk.reshape(-1, 16)
    now = max(tile_now(t) for t in T); new = max(tile2(t) for t in T)
    crit = T[np.argmax([tile_now(t) for t in T])]; s = np.sort(crit)[::-1]
    print(f"key {j}: now {now/GHZ/1e6:6.2f} -> {new/GHZ/1e6:6.2f} ms | critical tile: attempts with 13-16 live {s[12]:.0f}, 10-12 live {s[9]-s[12]:.0f}, 9 live {s[8]-s[9]:.0f}, 4-8 {s[3]-s[8]:.0f}, 3 {s[2]-s[3]:.0f}, <=2 {s[0]-s[2]:.0f}")
