"""knossos.competition -- (analysis model history): Knossos races :linear
against :wgl on two threads and returns whichever finishes first, so its
:analyzer varies run to run.  Here the choice is made up front (TBC_ALG_COMPETITION):
the level sweep (:linear) where it is the faster engine -- one history or a small
batch, no witness asked for -- handing what it cannot finish to the depth-first
search (:wgl), and the depth-first search for big batches; :analyzer says who answered."""
from . import _analysis


def analysis(model, history, opts=None):
    return _analysis.analysis(model, history, None, **(opts or {}))
