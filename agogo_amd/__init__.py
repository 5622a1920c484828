"""agogo_amd — MI355X-native self-play/MCTS hot path for gorgonia/agogo (libagz.so + thin bindings).

The product is the C-ABI shared library built from agogo_amd/csrc (see include/agz.h).  This Python
package only binds it (ctypes) for tests and bench.py; there is no CPU fallback: without the HIP
library and a GPU every compute entry point raises.
"""
from .capi import (AgzError, Arena, Comm, Ctx, Examples, GameConf, Mcts, MctsConf, Net, NetConf, State, Trainer, lib, lib_path,  # noqa: F401
                   rotate_boards, wino_h2_tile, wino_stages)
