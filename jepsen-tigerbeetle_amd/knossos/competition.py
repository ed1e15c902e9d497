"""knossos.competition -- (analysis model history): Knossos races :linear
against :wgl on two threads and returns whichever finishes first, so its
:analyzer varies run to run.  There is one device search here; it answers as
:wgl."""
from . import _analysis


def analysis(model, history, opts=None):
    return _analysis.analysis(model, history, None, **(opts or {}))
