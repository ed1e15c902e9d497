"""knossos.linear -- (analysis model history).

Knossos's linear analyzer sweeps the history keeping the set of reachable
configs (Lowe's just-in-time linearization).  Here that sweep is the kernel
jit_sweep.hip: the config set lives in LDS, the history is cut into segments
that are swept concurrently and composed (TBC_ALG_LINEAR in the C-ABI).  A
history whose config set outgrows on-chip memory is answered by the depth-first
search instead and says so in :analyzer."""
from . import _analysis


def analysis(model, history, opts=None):
    return _analysis.analysis(model, history, "linear", **(opts or {}))
