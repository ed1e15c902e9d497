"""Python mirror of the jepsen namespaces either side of the hot path:
jepsen.checker (Checker, linearizable, compose) and jepsen.independent
(tuple, checker) -- the call sites the reference composes its checkers with
(/root/reference/src/tigerbeetle/core.clj:139-146,
workloads/set_full.clj:155-158, tests/ledger.clj:363-367)."""
from . import checker, edn, independent  # noqa: F401
