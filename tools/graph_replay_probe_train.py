"""One fresh process: capture a training step (macx.CapturedTrainStep, verify off), replay it six times on new inputs and
compare every gradient bit for bit with the eager step.  Run it in many fresh processes -- the round-3 bug (a memset ordered
against atomics) failed in about one process in ten, from the second replay on:
    for i in $(seq 20); do python tools/graph_replay_probe_train.py | tail -1; done"""
import os, sys, torch
sys.path.insert(0, '.')
import macx
dev = torch.device('cuda:0')
B, S, N, d, p = 6, 7, 40, int(os.environ.get('DBG_D', '128')), 3
cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
step = macx.CapturedTrainStep(cfg, params, B, S, N, seed=11, verify=False)
bad = 0
for trial in range(6):
    vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=trial)]
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(trial)).to(dev)
    step.load(vq, words, lengths, kb, gm)
    mem = step.replay().clone()
    got = [t.grad.clone() for t in step._leaves()]
    keep = [t.grad for t in step._leaves()]
    ref_mem = step._eager().clone()
    ref = [t.grad.clone() for t in step._leaves()]
    for t, g in zip(step._leaves(), keep):
        t.grad = g
    torch.cuda.synchronize()
    bad += int(not (torch.equal(mem, ref_mem) and all(torch.equal(a, b) for a, b in zip(got, ref))))
print("train-step capture: bad replays", bad)
