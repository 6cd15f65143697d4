"""falsify or confirm the power-budget explanation of "a faster kernel does not make a faster step"
(profiles/r05_power_coupling.txt).  (GPU box)

The replayed training step with an IDLE GAP of G microseconds (one spinning thread, torch.cuda._sleep: no memory traffic, one
wave on one CU) inserted right after each grouped 3x3 weight gradient (5 per step, the largest kernels of the step), for
G in --gaps.  If the matrix kernels share an ENERGY window longer than a kernel, the kernels behind a gap find budget the gap
did not spend and run faster: busy time = step - 5 G shrinks with G (by about what round 5's faster weight gradient cost its
neighbours, 1-3 % of them).  If the coupling is cache state (what a shorter kernel leaves in L2 / MALL), an idle gap changes
nothing: busy time stays.  Alternating rounds on one box; `--trace` runs ONE variant for rocprofv3 --kernel-trace (per-kernel
durations with and without gaps come from two traced runs: tools/power_idle_ab.sh).
    python tools/power_idle_ab.py [--gaps 0,30,60,120] [--rounds 4] [--steps 150]
    python tools/power_idle_ab.py --trace 60 --steps 30"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd.engine import ResNet18Engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaps", default="0,30,60,120")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--trace", type=float, default=None, help="one variant only (gap in us), for a profiler run")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, S = 256, 224
    g = torch.Generator().manual_seed(1000)
    x = torch.randn(B, 3, S, S, generator=g).to(dev)
    y = torch.randint(0, 3, (B,), generator=g).to(dev)

    # cycles of torch.cuda._sleep per microsecond, measured
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)

    def build(gap_us):
        eng = ResNet18Engine(B, 3, 3, S, "max", dtype=torch.bfloat16, device=dev)
        eng.fuse_sgd_tail = True
        torch.manual_seed(42)
        eng.init_weights()
        n = int(gap_us * cyc_per_us)
        count = [0]
        if n > 0:
            def hook():
                count[0] += 1
                torch.cuda._sleep(n)
            eng.after_wgrad_hook = hook

        def step():
            eng.forward(x); eng.loss_backward(y); eng.sgd_step(1e-4, 5e-4)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        count[0] = 0
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            step()
        return eng, gph, count[0]

    def time_graph(gph, steps):
        for _ in range(10):
            gph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    if a.trace is not None:
        eng, gph, ngap = build(a.trace)
        ms = time_graph(gph, a.steps)
        print(json.dumps({"gap_us": a.trace, "gaps_per_step": ngap, "ms_per_step": round(ms, 4), "sleep_cycles_per_us": round(cyc_per_us, 1)}))
        return
    gaps = [float(v) for v in a.gaps.split(",")]
    built = {gp: build(gp) for gp in gaps}
    # the gap as the hardware runs it (spin + launch latency of the node), measured in a graph of nothing else
    gap_real = {}
    for gp in gaps:
        if gp <= 0:
            gap_real[gp] = 0.0
            continue
        n = int(gp * cyc_per_us)
        gg = torch.cuda.CUDAGraph()
        torch.cuda._sleep(n); torch.cuda.synchronize()
        with torch.cuda.graph(gg):
            for _ in range(50):
                torch.cuda._sleep(n)
        gap_real[gp] = time_graph(gg, 20) * 1e3 / 50
    res = {gp: [] for gp in gaps}
    for r in range(a.rounds):
        for gp in (gaps if r % 2 == 0 else gaps[::-1]):
            res[gp].append(time_graph(built[gp][1], a.steps))
    base = sorted(res[gaps[0]])[len(res[gaps[0]]) // 2] if gaps[0] == 0 else None
    print(f"sleep calibration: {cyc_per_us:.1f} cycles per us")
    print(f"{'gap us':>7s} {'gaps':>5s} {'gap as run us':>14s} {'step ms (median)':>17s} {'busy = step - gaps':>19s} {'busy vs gap 0':>14s}   rounds")
    for gp in gaps:
        v = sorted(res[gp]); med = v[len(v) // 2]
        ngap = built[gp][2]
        busy = med - ngap * gap_real[gp] * 1e-3
        rel = f"{(busy / base - 1) * 100:+.2f} %" if base else "-"
        print(f"{gp:7.0f} {ngap:5d} {gap_real[gp]:14.1f} {med:17.4f} {busy:19.4f} {rel:>14s}   " + " ".join(f"{t:.3f}" for t in res[gp]))


if __name__ == "__main__":
    main()
