"""knossos.wgl -- (analysis model history) and (analysis model history opts)."""
from . import _analysis


def analysis(model, history, opts=None):
    """Wing-Gong/Lowe search on the MI355X.  opts: {"time-limit": ms, ...}."""
    return _analysis.analysis(model, history, "wgl", **(opts or {}))
