"""Is the eager step slow on some boxes, or slow EARLY in a process?  The metric's step (B = 64, p = 12) in blocks of 20, with the
time since process start, for about 40 s."""
import os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, macx, bench
print("import done at %.1f s" % (time.perf_counter() - T0), flush=True)
dev = torch.device("cuda:0")


class _D:
    @staticmethod
    def get_backend():
        return "none"


step, params, kbd, bl = bench.make_step(macx, dev, _D, 1, 0, 64, 12, 1234)
i = 0
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(i)
        i += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("block %2d at %5.1f s: %.3f ms per step" % (blk, t0 - T0, (t1 - t0) / 20 * 1e3), flush=True)
    time.sleep(1.0)
