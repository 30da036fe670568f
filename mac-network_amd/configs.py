"""Host-side configuration helpers of the product package: the effective option sets of the reference's published flag
files (configs/args.txt, args1-4.txt over config.py defaults) as plain namespaces, and the synthetic CLEVR-shaped inputs
of SURVEY.md 8d.  `bench.py`, `__graft_entry__.smoke()` and user code build workloads from here; the test oracle keeps its
own copy so that nothing on the measured path imports `oracle/`.

    cfg = macx.configs.flag_file_config("args", netLength=12)
    vecQ, words, lengths, kb = macx.configs.synthetic_inputs(64, 50, 196, 512)
"""
from types import SimpleNamespace

import torch


def default_config(**over):
    """config.py defaults for every option the cell, the output unit, the stem and the encoder read."""
    c = SimpleNamespace(
        # dims / init (config.py:292-303)
        netLength=16, memDim=512, ctrlDim=512, attDim=512, unsharedCells=False,
        initCtrl="PRM", initMem="PRM", initKBwithQ="NON", addNullWord=False,
        # control (config.py:307-327)
        controlWholeQ=False, controlContinuous=False, controlContextual=False,
        controlInWordsProj=False, controlOutWordsProj=False, controlInputUnshared=False,
        controlInputAct="TANH", controlFeedPrev=False, controlFeedPrevAtt=False,
        controlFeedInputs=False, controlContAct="NON", controlConcatWords=False,
        controlProj=False, controlProjAct="NON",
        # read (config.py:344-362)
        readProjInputs=False, readProjShared=False, readMemAttType="MUL", readMemConcatKB=False,
        readMemConcatProj=False, readMemProj=False, readMemAct="RELU", readCtrl=False,
        readCtrlAttType="MUL", readCtrlConcatKB=False, readCtrlConcatProj=False,
        readCtrlConcatInter=False, readCtrlAct="RELU", readSmryKBProj=False,
        # write (config.py:369-387)
        writeInputs="BOTH", writeConcatMul=False, writeInfoProj=False, writeInfoAct="NON",
        writeSelfAtt=False, writeSelfAttMod="NON", writeMergeCtrl=False, writeMemProj=False,
        writeMemAct="NON", writeGate=False, writeGateShared=False, writeGateBias=1.0,
        # misc (config.py:194-223)
        memoryVariationalDropout=False, memoryDropout=0.85, readDropout=0.85, writeDropout=1.0,
        relu="STD", mulBias=0.0, memoryBN=False, bnDecay=0.999, bnCenter=False, bnScale=False,
        # output unit / classifier (model.py:512-576)
        outQuestion=False, outQuestionMul=False, outClassifierDims=[512], outputDropout=0.85,
        answerWordsNum=28,
        # question encoder (config.py:178-206, 262-270)
        wrdEmbDim=300, encDim=512, encType="LSTM", encBi=False, encNumLayers=1, encVariationalDropout=False,
        encProj=False, encProjQAct="NON", encInputDropout=0.85, qDropout=0.92, wrdEmbFixed=False,
    )
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError("unknown config flag %r" % k)
        setattr(c, k, v)
    return c


_COMMON = dict(memoryVariationalDropout=True, relu="ELU", outQuestion=True, controlContextual=True, encBi=True,
               readProjInputs=True, readMemConcatKB=True, readMemConcatProj=True, readMemProj=True,
               readCtrl=True, writeMemProj=True)
FLAG_FILES = {
    "args": dict(_COMMON, initCtrl="Q", controlInputUnshared=True),                                   # configs/args.txt
    "args1": dict(_COMMON, initCtrl="PRM", controlFeedPrev=True, controlFeedPrevAtt=True,
                  controlFeedInputs=True, controlContAct="TANH"),                                      # configs/args1.txt
    "args2": dict(_COMMON, initCtrl="Q", controlInputUnshared=True),                                  # configs/args2.txt
    "args3": dict(_COMMON, initCtrl="Q", controlInputUnshared=True, writeSelfAtt=True, writeSelfAttMod="CONT"),   # args3.txt
    "args4": dict(_COMMON, initCtrl="Q", controlInputUnshared=True, writeGate=True),                  # configs/args4.txt
}


def flag_file_config(name, **over):
    """Effective options of one of the reference's flag files, with keyword overrides (e.g. netLength=12)."""
    opts = dict(FLAG_FILES[name])
    opts.update(over)
    return default_config(**opts)


def synthetic_inputs(B, S, N, d, seed=1234, dtype=torch.float32):
    """SURVEY.md 8d cell-level inputs: vecQuestions, questionCntxWords ~ U(-1, 1) (rows past the question length zeroed),
    questionLengths ~ randint[3, S] with question 0 at full length, knowledgeBase = ELU(N(0, 1)) (the stem's last op)."""
    g = torch.Generator().manual_seed(seed)
    vecQ = (torch.rand((B, d), generator=g, dtype=torch.float64) * 2 - 1)
    words = (torch.rand((B, S, d), generator=g, dtype=torch.float64) * 2 - 1)
    lengths = torch.randint(3, S + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = S
    m = (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.float64)
    words = words * m.unsqueeze(-1)
    kb = torch.nn.functional.elu(torch.randn((B, N, d), generator=g, dtype=torch.float64))
    return vecQ.to(dtype), words.to(dtype), lengths, kb.to(dtype)
