import os, sys, torch, warnings
sys.path.insert(0, '.')
import macx
warnings.simplefilter("ignore")
dev = torch.device('cuda:0')
B, S, N, d, p = int(os.environ.get("PB", 64)), 50, 196, 512, int(os.environ.get("PP", 12))
cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
step = macx.CapturedTrainStep(cfg, params, B, S, N, seed=5, warmup=int(os.environ.get("PW", 2)))
names = ["memory", "vecQ", "words", "kb"] + list(params.fields)
print("captured", step.captured, [(r, names[i]) for r, i in step.verify_report])
