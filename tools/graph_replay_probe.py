"""Probe for the graph-replay mismatch CapturedForward guards against (mac-network_amd/graph.py; fixed at its root in absmax4,
round 3: 0 of 20 processes fail now, 1 in 3 did): one capture, six replays, each
compared bit for bit with the eager run.  Run it in several fresh processes -- a process either replays correctly every time
or is wrong from its SECOND replay on:
    for i in 1 2 3 4 5 6 7 8; do DBG=noeager python tools/graph_replay_probe.py | tail -1; done
DBG=plain  an eager run between replays;  DBG=zero / DBG=nan  fill the captured run's `saved` buffer before every replay
(neither changes the outcome: stale contents of `saved` are not what the bad replays read);  DBG_D=<width>, DBG_CHAIN=0  other
widths / the per-product kernels instead of the chain kernels (both fail at the same rate)."""
import os, sys, torch
sys.path.insert(0, '.')
import macx
dev = torch.device('cuda:0')
B, S, N, d, p = 6, 7, 40, int(os.environ.get('DBG_D', '128')), 3
if os.environ.get('DBG_CHAIN'): macx._lib.lib().macx_debug_set(4, int(os.environ['DBG_CHAIN']))
cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
cap = macx.CapturedForward(cfg, params, B, S, N, verify=False)
def eager(vq, words, lengths, kb):
    with torch.no_grad():
        cell = macx.MACCell(vq, words, words, lengths, kb, 1.0, 1.0, 1.0, B, False, config=cfg, params=params)
        return cell.run().memory.clone()
vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=1)]
base = eager(vq, words, lengths, kb)
mode = os.environ.get("DBG", "plain")
bad = 0
for trial in range(6):
    if mode == "plain":
        a = eager(vq, words, lengths, kb)
    if mode == "zero":
        cap.cell._run.saved.zero_()
    if mode == "nan":
        cap.cell._run.saved.fill_(float("nan"))
    g = cap(vq, words, lengths, kb).clone(); torch.cuda.synchronize()
    bad += int(not torch.equal(g, base))
    if mode == "nan" and not torch.isfinite(g).all(): print("non-finite output at trial", trial)
print("mode", mode, "bad replays", bad)
