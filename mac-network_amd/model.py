"""The device-side graph of MACnet.build (model.py:774-824) from the stem to the logits, with the
question encoder's outputs as inputs:

    images --stem--> knowledgeBase --MACCell x netLength--> memory --outputOp/classifier--> logits

`MACNetCore` takes the question encoder's outputs (vecQuestions / questionCntxWords / questionLengths) as inputs;
`MACNet` adds `embeddingsOp` + `encoder` (model.py:208-307) in front, so its inputs are exactly
the reference's feed dict: question word ids, question lengths, image features, (answers).  Out of this graph stay
only the host side (preprocess.py / main.py) and the baseline / unused stem variants."""
import torch

from .cell import MACCell
from .encoder import QuestionEncoder
from .options import UnsupportedOptions, freeze, fresh_seed, get
from .output import OutputClassifier, answer_loss_and_pred
from .params import MACCellParams
from .stem import Stem


class MACNetCore(torch.nn.Module):
    def __init__(self, config, H=14, W=14, imageInDim=1024, answerWordsNum=28, generator=None):
        super().__init__()
        self.config = config
        self.netLength = int(get(config, "netLength"))
        self.stem = Stem(config, H=H, W=W, inDim=imageInDim, generator=generator)
        try:
            freeze(config)
            self.cell = MACCellParams(config, self.netLength, generator=generator)
        except UnsupportedOptions:
            # an option set of the generic path (the reference's default configuration among them).  The option set and the
            # net length are known here, so the plan is compiled now and every variable it names is created now, in the plan's
            # (= TensorFlow's creation) order: tensors(), the optimizer's layout and a checkpoint load (missing / extra / shape
            # checks of checkpoint.load_reference) see exactly the variables the cell will use -- nothing is adopted by scope and
            # nothing is drawn at random behind a loaded checkpoint's back on the first forward pass
            from . import plan as _plan
            from .generic import GenericParams
            self.cell = GenericParams(generator=generator)
            for name, spec in _plan.compile_cell(config, self.netLength).variables.items():
                self.cell.ensure(name, spec.shape, spec.init)
        self.out = OutputClassifier(config, answerWordsNum=answerWordsNum, generator=generator)

    def tensors(self):
        return self.stem.tensors() + self.cell.tensors() + self.out.tensors()

    def forward(self, images, vecQuestions, questionCntxWords, questionLengths, train=False, seed=None, b0=0,
                questionWords=None):
        cfg = self.config
        if questionWords is None:
            if not get(cfg, "controlContextual"):
                # mac_cell.py:570 would select the raw word embeddings; silently attending over the encoder outputs instead
                # would diverge from the reference
                raise UnsupportedOptions("without --controlContextual the cell attends over the raw word embeddings "
                                         "(mac_cell.py:570): pass them as questionWords=[B,S,ctrlDim]")
            questionWords = questionCntxWords
        seed = fresh_seed(seed, train)
        kb = self.stem(images, train=train, seed=seed, b0=b0)                       # model.py:791
        cell = MACCell(vecQuestions=vecQuestions, questionWords=questionWords, questionCntxWords=questionCntxWords,
                       questionLengths=questionLengths, knowledgeBase=kb, memoryDropout=get(cfg, "memoryDropout"),
                       readDropout=get(cfg, "readDropout"), writeDropout=get(cfg, "writeDropout"), batchSize=images.shape[0],
                       train=train, config=cfg, params=self.cell, netLength=self.netLength, seed=seed, b0=b0)
        state = cell.run()                                                           # model.py:801 (MACnetwork)
        self.last_cell = cell
        return self.out(state.memory, vecQuestions, train=train, seed=seed, b0=b0)   # model.py:805-809

    @staticmethod
    def loss_and_pred(logits, answers):
        return answer_loss_and_pred(logits, answers)                                 # model.py:812-813


class MACNet(MACNetCore):
    """MACnet.build's tower body (model.py:781-813): embeddingsOp -> encoder -> stem -> MACnetwork -> outputOp/classifier."""

    def __init__(self, config, vocab, H=14, W=14, imageInDim=1024, answerWordsNum=28, embInit=None, generator=None):
        super().__init__(config, H=H, W=W, imageInDim=imageInDim, answerWordsNum=answerWordsNum, generator=generator)
        self.enc = QuestionEncoder(config, vocab, embInit=embInit, generator=generator)

    def tensors(self):
        return self.enc.tensors() + super().tensors()

    def forward(self, images, questions, questionLengths, train=False, seed=None, b0=0, check_ids=True):
        seed = fresh_seed(seed, train)
        words, vecQ = self.enc(questions, questionLengths, train=train, seed=seed, b0=b0, check_ids=check_ids)   # model.py:783-788
        raw = None
        if not get(self.config, "controlContextual"):
            # mac_cell.py:570: the control unit attends over the embedded words themselves (embeddingsOp's output, no dropout)
            raw = self.enc.embed(questions)
            if raw.shape[-1] != get(self.config, "ctrlDim"):
                raise ValueError("Dimensions must be equal: without --controlContextual the question words are wrdEmbDim = %d wide, "
                                 "the control state ctrlDim = %d (mac_cell.py:154)" % (raw.shape[-1], get(self.config, "ctrlDim")))
        return super().forward(images, vecQ, words, questionLengths, train=train, seed=seed, b0=b0, questionWords=raw)
