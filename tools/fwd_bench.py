import sys, time, torch
sys.path.insert(0, "/root/repo")
import macx
dev = torch.device("cuda:0")
B, S, N, D = 64, 50, 196, 512
for P in (4, 8, 12, 16, 4):
    cfg = macx.configs.flag_file_config("args", netLength=P)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D)
    params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1)).to(dev)
    vqd, wd, kbd, ld = vq.to(dev), words.to(dev), kb.to(dev), lengths.to(dev)
    def fwd():
        with torch.no_grad():
            cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kbd,
                                memoryDropout=1.0, readDropout=1.0, writeDropout=1.0, batchSize=B, train=False, config=cfg, params=params)
            return cell.run().memory
    for _ in range(5): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fwd()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("forward-only p=%d B=%d: %.3f ms/batch  %.0f questions/s" % (P, B, dt * 1e3, B / dt))
