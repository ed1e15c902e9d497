"""knossos.op -- predicates on an op map's :type.

The reference uses exactly these: op/invoke? op/ok? op/fail?
(/root/reference/src/tigerbeetle/tests/ledger.clj:165,212,231;
workloads/set_full.clj:58,64; checker/perf.clj:108,115,124,133).
Ops are dicts with keys type, f, value, process, index, time; keyword values
are plain strings ("invoke", "ok", "fail", "info", "read", ...).
"""


def invoke(process, f, value):
    return {"type": "invoke", "f": f, "value": value, "process": process}


def ok(process, f, value):
    return {"type": "ok", "f": f, "value": value, "process": process}


def fail(process, f, value):
    return {"type": "fail", "f": f, "value": value, "process": process}


def info(process, f, value):
    return {"type": "info", "f": f, "value": value, "process": process}


def invoke_p(op):
    return op["type"] == "invoke"


def ok_p(op):
    return op["type"] == "ok"


def fail_p(op):
    return op["type"] == "fail"


def info_p(op):
    return op["type"] == "info"
