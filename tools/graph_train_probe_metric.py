"""capture the metric-shape training step, replay, and name the tensors that differ from the eager step"""
import os, sys, torch
sys.path.insert(0, '.')
import macx
dev = torch.device('cuda:0')
B, S, N, d, p = int(os.environ.get("PB", 64)), 50, int(os.environ.get("PN", 196)), int(os.environ.get("PD", 512)), int(os.environ.get("PP", 12))
cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
step = macx.CapturedTrainStep(cfg, params, B, S, N, seed=5, verify=False)
names = ["vecQ", "words", "kb"] + list(params.fields)
g = torch.Generator().manual_seed(20240520)
with torch.no_grad():
    for t in (step.vecQuestions, step.words, step.knowledgeBase, step.d_memory):
        t.copy_(torch.randn(t.shape, generator=g).to(dev))
cap = [t.grad for t in step._leaves()]
mem = step._eager().clone()
want = [t.grad.clone() for t in step._leaves()]
mem2 = step._eager().clone()
want2 = [t.grad.clone() for t in step._leaves()]
print("eager vs eager differ:", [n for n, a, b in zip(names, want, want2) if not torch.equal(a, b)], torch.equal(mem, mem2))
for t, gc in zip(step._leaves(), cap):
    t.grad = gc
for r in range(3):
    step.graph.replay(); torch.cuda.synchronize()
    bad = [(n, float((t.grad - w).abs().max() / w.abs().max())) for n, t, w in zip(names, step._leaves(), want) if not torch.equal(t.grad, w)]
    print("replay", r, "memory equal:", torch.equal(step.memory, mem), " differing:", bad)
