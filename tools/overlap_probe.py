"""How much would a lone 2^24 MSM gain if its memory-bound front end (recode / partition / sort, 2.0 ms) and its small back end (combine /
reduce, 0.7 ms) ran beside the VALU-bound accumulation of another part of the same MSM?  Upper bound measured without touching the
pipeline: TWO whole MSMs on two contexts of the same device (own stream, own scratch, one thread each) against the same two MSMs one
after the other.  If the pair does not finish sooner than 2 x the lone time, splitting one MSM's windows over two streams cannot gain either.
    python tools/overlap_probe.py [log_n] [rounds]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tiny_ram_halo2_amd import api, synth


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = 1 << log_n
    api.init(0)
    sc = synth.field_elements(0x0F1A + log_n, n)
    want = None
    lanes = []
    for k in range(2):
        ctx = api.Context(0)
        with ctx:
            b = api.Bases.generate("pallas", synth.BASE_S0, synth.BASE_D, n)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                d = torch.from_numpy(sc.view(np.int64)).cuda()
            s.synchronize()
            r = b.msm_dev(d, n, stream=s.cuda_stream)
            r = b.msm_dev(d, n, stream=s.cuda_stream)
            if want is None:
                want = r
            assert (r == want).all()
        lanes.append((ctx, b, s, d))

    def run(lane, count, barrier=None):
        ctx, b, s, d = lane
        with ctx:
            if barrier is not None:
                barrier.wait()
            for _ in range(count):
                r = b.msm_dev(d, n, stream=s.cuda_stream)
            assert (r == want).all()

    t = time.perf_counter()
    run(lanes[0], rounds)
    lone = (time.perf_counter() - t) / rounds
    t = time.perf_counter()
    run(lanes[1], rounds)
    lone2 = (time.perf_counter() - t) / rounds
    bar = threading.Barrier(3)
    th = [threading.Thread(target=run, args=(lanes[k], rounds, bar)) for k in range(2)]
    for x in th:
        x.start()
    bar.wait()
    t = time.perf_counter()
    for x in th:
        x.join()
    pair = (time.perf_counter() - t) / rounds
    print(f"2^{log_n} Pallas MSM: lone {lone * 1e3:.2f} / {lone2 * 1e3:.2f} ms; two at once on two contexts: {pair * 1e3:.2f} ms per pair "
          f"= {pair / (lone + lone2):.3f} of the two lone times ({2 * n / pair / 1e6:.0f} M pairs/s against {n / lone / 1e6:.0f})")


if __name__ == "__main__":
    main()
