"""Does a leg of bench.py run slower BEHIND another one in the same process?  argv: leg names in order, e.g.
   python tools/leg_order_probe.py b128 | graph b128 | fwd b128 | graph fwd b128 model"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, macx, bench
import torch.distributed as dist
dev = torch.device("cuda:0")


class _D:
    @staticmethod
    def get_backend():
        return "none"


for leg in sys.argv[1:]:
    t0 = time.perf_counter()
    if leg == "b128":
        r = bench.train_b128_p12(macx, dev, _D, 1234)
    elif leg == "graph":
        r = bench.train_step_graph(macx, dev, 1234)
    elif leg == "fwd":
        r = bench.fwd_only_p4(macx, dev, 1234)
    elif leg == "model":
        r = bench.model_level(macx, dev, 1234)
    elif leg == "gqa":
        r = bench.gqa_shape_p4(macx, dev, 1234, "args3")
    elif leg == "empty":
        torch.cuda.empty_cache(); r = {"ms_per_step": 0}
    ms = torch.cuda.memory_stats()
    print("%-6s %8.3f ms  (leg took %.1f s; allocator: device_alloc %d device_free %d reserved %.1f GB)" % (
        leg, r.get("ms_per_step", r.get("ms_per_batch", 0)), time.perf_counter() - t0, ms.get("num_device_alloc", -1), ms.get("num_device_free", -1),
        ms.get("reserved_bytes.all.current", 0) / 2**30), flush=True)
