"""knossos.history -- the helpers Knossos applies before a search and the ones
the reference calls directly: unmatched-invokes
(/root/reference/src/tigerbeetle/tests/ledger.clj:206), pair-index+ and
completion (checker/perf.clj:617,623), plus index / complete /
without-failures (recalled).  Pure host code over lists of op
dicts; the columnar equivalent that feeds the GPU is tbc_pair_events."""
from __future__ import annotations


def client_op(op):
    """The reference filters on (int? process): tests/ledger.clj:94,204,228."""
    return isinstance(op.get("process"), int) and not isinstance(op.get("process"), bool)


def index(history):
    """Assign :index = position (ensure-indexed)."""
    out = []
    for i, op in enumerate(history):
        if op.get("index") != i:
            op = dict(op, index=i)
        out.append(op)
    return out


def pair_index(history):
    """{index of invocation: index of its completion, and back}; unmatched invokes map to None."""
    pairs, open_by_proc = {}, {}
    for i, op in enumerate(history):
        if not client_op(op):
            continue
        p = op["process"]
        if op["type"] == "invoke":
            open_by_proc[p] = i
            pairs[i] = None
        elif p in open_by_proc:
            j = open_by_proc.pop(p)
            pairs[j] = i
            pairs[i] = j
    return pairs


def completion(history, pairs, i):
    j = pairs.get(i)
    return None if j is None else history[j]


def unmatched_invokes(history):
    pairs = pair_index(history)
    return [history[i] for i, j in pairs.items() if j is None and history[i]["type"] == "invoke"]


def complete(history):
    """Fold each :ok completion's :value back into its invocation; mark invocations
    that :fail with fails? True."""
    history = [dict(op) for op in history]
    pairs = pair_index(history)
    for i, op in enumerate(history):
        if client_op(op) and op["type"] == "invoke":
            j = pairs.get(i)
            if j is not None:
                if history[j]["type"] == "ok":
                    op["value"] = history[j]["value"]
                elif history[j]["type"] == "fail":
                    op["fails?"] = True
                    history[j]["fails?"] = True
    return history


def without_failures(history):
    """Drop invocations that failed and their :fail completions (they did not happen)."""
    return [op for op in history if not op.get("fails?") and op["type"] != "fail"]
