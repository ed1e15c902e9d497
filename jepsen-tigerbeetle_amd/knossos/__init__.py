"""Python mirror of the knossos namespaces on (or next to) the hot path:
knossos.op, knossos.history, knossos.model, knossos.model.memo, knossos.wgl,
knossos.linear, knossos.competition -- same names, argument meaning and result
maps (Knossos itself is not in /root/reference)."""
from . import op, history, model, memo, wgl, linear, competition  # noqa: F401
