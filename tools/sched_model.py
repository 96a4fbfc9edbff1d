#!/usr/bin/env python3
"""Wave-scheduling model of render_kernel_stream (dev tool; no GPU needed).

Input: the joint histogram (inner-node steps x leaf tests per trace) a -DRT_PHASE_TIMERS build writes with
RT06_TRACE_HIST=file (one MI355X run of config 2).  The model replays one 64-lane wave under a scheduling policy with
per-phase issue costs taken from the phase timers (unit = one hot-loop step) and reports lane utilisation per phase and
the cost per trace.  Used to decide which scheduling changes are worth building.

    python tools/sched_model.py gpurun_out/trace_hist_book1.txt [--policy one|two] [--keep 40 --shade 56 --leaf 4 --swap 16]
"""
import argparse, random, sys
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("hist")
ap.add_argument("--policy", default="one", choices=["one", "two"])
ap.add_argument("--keep", type=int, default=40)
ap.add_argument("--shade", type=int, default=56)
ap.add_argument("--leaf", type=int, default=4)
ap.add_argument("--swap", type=int, default=16)
ap.add_argument("--stall", type=int, default=8)
ap.add_argument("--traces", type=int, default=200000)
ap.add_argument("--c-leaf", type=float, default=1.2)     # one leaf phase, in hot steps
ap.add_argument("--c-outer", type=float, default=0.25)   # outer-loop / phase-check overhead per iteration
ap.add_argument("--c-shade", type=float, default=7.0)    # shade body of one round
ap.add_argument("--c-regen", type=float, default=1.7)    # regeneration part of a round
ap.add_argument("--c-begin", type=float, default=2.0)    # begin-trace part of a round (or of a swap phase)
ap.add_argument("--c-swap", type=float, default=0.4)     # register moves + path-state traffic of a swap phase
ap.add_argument("--first-leaf", type=int, default=8)    # inner steps before the first leaf can appear
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rnd = random.Random(a.seed)

h = np.loadtxt(a.hist)
flat = (h / h.sum()).ravel()
cum = np.cumsum(flat)

def new_trace():
    """event list of one trace: 'I' inner step / 'L' leaf test"""
    k = int(np.searchsorted(cum, rnd.random()))
    H, L = divmod(k, h.shape[1])
    ev = ['I'] * H
    # a trace walks ~8 levels down before it can meet its first leaf (488 leaves, median split: leaves sit at depth 8-9), so
    # the leaf tests fall into the later part of the trace
    first = min(H, a.first_leaf)
    for _ in range(L):
        ev.insert(rnd.randint(first, len(ev)), 'L')
    return ev

class Lane:
    __slots__ = ("ev", "pos", "parked")   # parked: None (empty) / 'W' (waits for shade) / 'R' (ready to trace)
    def __init__(self):
        self.ev, self.pos, self.parked = new_trace(), 0, None
    def state(self):
        return 'S' if self.pos >= len(self.ev) else self.ev[self.pos]

lanes = [Lane() for _ in range(64)]
t = 0.0
acc = dict(hot=0.0, hot_lanes=0.0, leaf=0.0, leaf_lanes=0.0, shade=0.0, shade_lanes=0.0, swap=0.0, swap_lanes=0.0, outer=0.0)
done = 0
rounds = 0
while done < a.traces:
    st = [l.state() for l in lanes]
    n_inner = st.count('I')
    # phase 1
    if n_inner:
        while True:
            k = 0
            for l in lanes:
                if l.state() == 'I':
                    l.pos += 1; k += 1
            acc["hot"] += 1.0; acc["hot_lanes"] += k
            n_inner = sum(1 for l in lanes if l.state() == 'I')
            if n_inner < a.keep: break
    # phase 2
    n_leaf = sum(1 for l in lanes if l.state() == 'L')
    n_inner = sum(1 for l in lanes if l.state() == 'I')
    if n_leaf and (n_leaf >= a.leaf or n_inner == 0):
        for l in lanes:
            if l.state() == 'L': l.pos += 1
        acc["leaf"] += a.c_leaf; acc["leaf_lanes"] += n_leaf * a.c_leaf
    acc["outer"] += a.c_outer
    n_trav = sum(1 for l in lanes if l.state() != 'S')
    if a.policy == "one":
        if 64 - n_trav >= a.shade or n_trav == 0:
            k = 64 - n_trav
            c = a.c_shade + a.c_regen + a.c_begin
            acc["shade"] += c; acc["shade_lanes"] += k * c
            for l in lanes:
                if l.state() == 'S':
                    l.ev, l.pos = new_trace(), 0
                    done += 1
            rounds += 1
    else:
        # swap phase: active finished and the parked slot can take it (empty) or offers a ready ray
        can_swap = [l for l in lanes if l.state() == 'S' and l.parked != 'W']
        if can_swap and (len(can_swap) >= a.swap or n_trav == 0):
            c = a.c_begin + a.c_swap
            acc["swap"] += c; acc["swap_lanes"] += len(can_swap) * c
            for l in can_swap:
                l.parked = 'W'               # the finished trace waits for its shade in the parked slot
                l.ev, l.pos = new_trace(), 0  # the ready ray (or a fresh primary ray) becomes the active one
            n_trav = sum(1 for l in lanes if l.state() != 'S')
        n_wait = sum(1 for l in lanes if l.parked == 'W')
        n_stall = sum(1 for l in lanes if l.state() == 'S' and l.parked == 'W')
        if n_wait and (n_wait >= a.shade or n_stall >= a.stall or n_trav == 0):
            c = a.c_shade + a.c_regen
            acc["shade"] += c; acc["shade_lanes"] += n_wait * c
            for l in lanes:
                if l.parked == 'W':
                    l.parked = 'R'
                    done += 1
            rounds += 1

tot = acc["hot"] + acc["leaf"] + acc["shade"] + acc["swap"] + acc["outer"]
print(f"policy {a.policy}: cost per trace {tot / done * 64:.2f} hot-step units per 64 traces "
      f"(hot {acc['hot']/tot:.1%}, leaf {acc['leaf']/tot:.1%}, shade {acc['shade']/tot:.1%}, swap {acc['swap']/tot:.1%}, outer {acc['outer']/tot:.1%})")
print(f"  lane utilisation: hot {acc['hot_lanes']/acc['hot']/64:.1%}, leaf {acc['leaf_lanes']/max(acc['leaf'],1e-9)/64:.1%}, "
      f"shade {acc['shade_lanes']/acc['shade']/64:.1%}, swap {acc['swap_lanes']/max(acc['swap'],1e-9)/64:.1%}; "
      f"hot steps per shade round {acc['hot']/rounds:.1f}, traces per round {done/rounds:.1f}")
