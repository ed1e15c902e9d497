"""knossos.linear -- (analysis model history).

Knossos's linear analyzer sweeps the history keeping the set of reachable
configs (Lowe's just-in-time linearization).  Its verdict and its failure
report (:op = first completion no config can pass, :previous-ok) are
properties of (model, history), not of the sweep, and the WGL kernel computes
exactly those; so this entry point answers from the same device search and
labels the result :analyzer :linear."""
from . import _analysis


def analysis(model, history, opts=None):
    r = _analysis.analysis(model, history, "linear", **(opts or {}))
    r["analyzer"] = "linear"
    return r
