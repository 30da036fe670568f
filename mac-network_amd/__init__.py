"""mac-network_amd: an MI355X-native MAC reasoning cell (control / read / write units of
stanfordnlp/mac-network's mac_cell.py and the ops.py primitives they call) behind the
reference's own MACCell interface.  Import with `importlib.import_module("mac-network_amd")`
or through the `macx` alias module at the repo root."""
from . import _lib, build, cell, checkpoint, configs, dp, encoder, generic, graph, h5, optim, options, output, params, plan, stem, tf_bundle   # noqa: F401
from .cell import MACCell, MACCellTuple             # noqa: F401
from .options import UnsupportedOptions, freeze     # noqa: F401
from .params import MACCellParams                   # noqa: F401
from .generic import GenericMACCell, GenericParams  # noqa: F401
from .output import GenericOutputClassifier, OutputClassifier   # noqa: F401
from .stem import Stem                              # noqa: F401
from .encoder import GenericQuestionEncoder, QuestionEncoder   # noqa: F401
from .model import MACNet, MACNetCore               # noqa: F401
from .graph import CapturedForward, CapturedTrainStep, CapturedDPTrainStep  # noqa: F401

__all__ = ["MACCell", "MACCellTuple", "MACCellParams", "OutputClassifier", "GenericOutputClassifier", "UnsupportedOptions", "freeze"]
